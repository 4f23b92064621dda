// Relative-pose arithmetic of pose.hip (fp64, one hypothesis per thread): the Nister five-point solver, Sampson error,
// essential-matrix decomposition and linear triangulation behind `estimate_pose` (/root/reference/src/utils/metrics.py:69-94,
// which delegates to cv2.findEssentialMat / cv2.recoverPose).  Plain C++ with no HIP intrinsics, so that the same functions
// are compiled for the device by pose.hip and for the host by tests/native/pose_host_check.cpp, where they are checked
// against oracle/pose_ref.py without a GPU.
#pragma once
#include <math.h>

#if defined(__HIPCC__)
#define POPE_HD __host__ __device__ inline __attribute__((always_inline))   // helpers taking array pointers must inline, or the
                                                                            // arrays they touch are pinned to scratch memory
#define POPE_UNROLL _Pragma("unroll")   // small fixed loops over thread-private arrays: keeps them in registers (no scratch)
#define POPE_ROLLED _Pragma("nounroll")  // a loop around a large inlined body that must stay one copy (instruction cache)
#else
#define POPE_HD inline __attribute__((always_inline))
#define POPE_UNROLL
#define POPE_ROLLED
#endif

namespace pose {

constexpr int ROUND = 256;          // hypotheses per round (= threads of the RANSAC workgroup; oracle/pose_ref.py:ROUND)

// ---- minimal-sample selection: counter-based, identical to oracle/pose_ref.py:sample_indices --------------------------
POPE_HD unsigned long long splitmix64(unsigned long long x) {
    x += 0x9E3779B97F4A7C15ull;
    x = (x ^ (x >> 30)) * 0xBF58476D1CE4E5B9ull;
    x = (x ^ (x >> 27)) * 0x94D049BB133111EBull;
    return x ^ (x >> 31);
}
POPE_HD void sample_indices(unsigned long long seed, unsigned h, unsigned n, int* picks /* [5] */) {
    for (int s = 0; s < 5; ++s) {
        for (unsigned a = 0;; ++a) {
            const unsigned long long key = seed ^ (h * 0xD1B54A32D192ED03ull) ^ ((unsigned long long)(s * 64 + a) * 0x8CB92BA72F3D8DD7ull);
            const int v = int(splitmix64(key) % n);
            bool dup = false;
            for (int k = 0; k < s; ++k) dup = dup || picks[k] == v;
            if (!dup) { picks[s] = v; break; }
        }
    }
}

// ---- polynomials in (x, y, z): linear [x, y, z, 1], quadratic [x2, y2, z2, xy, xz, yz, x, y, z, 1], cubic in Nister's
// elimination order [x3, y3, x2y, xy2, x2z, x2, y2z, y2, xyz, xy | xz2, xz, x, yz2, yz, y, z3, z2, z, 1] -----------------
POPE_HD int qidx(int a, int b) {
    const int t[4][4] = {{0, 3, 4, 6}, {3, 1, 5, 7}, {4, 5, 2, 8}, {6, 7, 8, 9}};
    return t[a][b];
}
POPE_HD int cidx(int q, int l) {
    const int t[10][4] = {{0, 2, 4, 5}, {3, 1, 6, 7}, {10, 13, 16, 17}, {2, 3, 8, 9}, {4, 8, 10, 11},
                          {8, 6, 13, 14}, {5, 9, 11, 12}, {9, 7, 14, 15}, {11, 14, 17, 18}, {12, 15, 18, 19}};
    return t[q][l];
}
// q += s * a * b  (a, b linear)
POPE_HD void mac_ll(double* q, const double* a, const double* b, double s) {
    for (int i = 0; i < 4; ++i)
        for (int j = 0; j < 4; ++j) q[qidx(i, j)] += s * a[i] * b[j];
}
// c += s * q * l  (q quadratic, l linear)
POPE_HD void mac_ql(double* c, const double* q, const double* l, double s) {
    for (int i = 0; i < 10; ++i)
        for (int j = 0; j < 4; ++j) c[cidx(i, j)] += s * q[i] * l[j];
}

// ---- real roots of a polynomial of degree <= 10 by Sturm sequences -----------------------------------------------------
// Every array below is indexed by unrolled loop counters only (bounds that depend on data are predicates, not trip counts):
// the chain then lives in registers on the device instead of scratch memory.  Polynomials are stored LEADING COEFFICIENT
// FIRST: g[m] = coefficient of t^(deg - m), zeros after position deg, so that a division step and Horner's rule touch
// positions that are sums of loop counters.
struct Sturm {
    double g[11][11];   // chain polynomial i has degree <= 10 - i
    int deg[11];
    int len;
};
// drop leading coefficients that vanished (|g[0]| <= tol) from a polynomial of degree <= MAXD
template <int MAXD>
POPE_HD void strip_leading(double* g, int& deg, double tol) {
    POPE_UNROLL
    for (int s = 0; s < MAXD; ++s) {
        const bool sh = deg > 0 && fabs(g[0]) <= tol;
        POPE_UNROLL
        for (int m = 0; m < MAXD; ++m) g[m] = sh ? g[m + 1] : g[m];
        g[MAXD] = sh ? 0.0 : g[MAXD];
        deg -= sh ? 1 : 0;
    }
}
template <int MAXD>
POPE_HD double horner_lead(const double* g, int deg, double t) {
    double v = g[0];
    POPE_UNROLL
    for (int m = 1; m <= MAXD; ++m) v = (m <= deg) ? v * t + g[m] : v;
    return v;
}
// link I of the chain: g[I] = -(g[I-2] mod g[I-1]), rescaled; returns false when the chain ends before it
template <int I>
POPE_HD bool sturm_link(Sturm& s) {
    constexpr int MA = 12 - I, MB = 11 - I;      // largest degrees g[I-2] and g[I-1] can have
    const int da = s.deg[I - 2], db = s.deg[I - 1];
    if (!(db > 0)) return false;
    const double* a = s.g[I - 2];
    const double* b = s.g[I - 1];
    const int dq = da - db;                       // >= 1
    double r[MA + 1];
    POPE_UNROLL
    for (int m = 0; m <= MA; ++m) r[m] = a[m];
    POPE_UNROLL
    for (int t = 0; t < MA; ++t) {                // long division r <- a mod b: dq + 1 steps (two, unless degrees dropped)
        if (t <= dq) {
            const double q = r[t] / b[0];
            POPE_UNROLL
            for (int j = 1; j <= MB; ++j)
                if (t + j <= MA) r[t + j] -= q * b[j];      // b is zero after position db: those steps change nothing
            r[t] = 0.0;
        }
    }
    double rm = 0.0, bm = 0.0;                    // the remainder sits at positions dq + 1 .. da, zeros everywhere else
    POPE_UNROLL
    for (int m = 0; m <= MA; ++m) { rm = fmax(rm, fabs(r[m])); bm = fmax(bm, fabs(a[m])); }
    if (!(rm > 1e-13 * bm)) return false;         // exact division: a and b share the remaining factor (multiple roots)
    const int sh = dq + 1;                        // move it to the front: shifts by 1, 2, 4, 8 as sh has the bits
    POPE_UNROLL
    for (int bit = 1; bit <= 8; bit *= 2) {
        const bool on = (sh & bit) != 0;
        POPE_UNROLL
        for (int m = 0; m <= MA; ++m) r[m] = on ? (m + bit <= MA ? r[m + bit] : 0.0) : r[m];
    }
    const double inv = -1.0 / rm;                 // negated remainder; the positive rescaling keeps the signs
    double* o = s.g[I];
    POPE_UNROLL
    for (int m = 0; m <= 10; ++m) o[m] = (m < MB) ? r[m] * inv : 0.0;
    int d = db - 1;
    strip_leading<MB - 1>(o, d, 1e-13);
    s.deg[I] = d;
    s.len = I + 1;
    return true;
}
// c[k] = coefficient of t^k, k <= d <= 10
POPE_HD void sturm_build(Sturm& s, const double* c, int d) {
    double mx = 0.0;
    for (int k = 0; k <= d; ++k) mx = fmax(mx, fabs(c[k]));
    POPE_UNROLL
    for (int i = 0; i <= 10; ++i) {
        s.deg[i] = 0;
        POPE_UNROLL
        for (int m = 0; m <= 10; ++m) s.g[i][m] = 0.0;
    }
    POPE_UNROLL
    for (int m = 0; m <= 10; ++m) s.g[0][m] = (m <= d && mx > 0.0) ? c[d - m] / mx : 0.0;
    int d0 = d;
    strip_leading<10>(s.g[0], d0, 1e-14);
    s.deg[0] = d0;
    s.len = 1;
    if (d0 == 0) return;
    POPE_UNROLL
    for (int m = 0; m <= 9; ++m) s.g[1][m] = (m < d0) ? double(d0 - m) * s.g[0][m] : 0.0;
    s.deg[1] = d0 - 1;
    s.len = 2;
    // each link ends the chain for good when it fails
    if (!sturm_link<2>(s)) return;
    if (!sturm_link<3>(s)) return;
    if (!sturm_link<4>(s)) return;
    if (!sturm_link<5>(s)) return;
    if (!sturm_link<6>(s)) return;
    if (!sturm_link<7>(s)) return;
    if (!sturm_link<8>(s)) return;
    if (!sturm_link<9>(s)) return;
    sturm_link<10>(s);
}
template <int I>
POPE_HD void sturm_sign(const Sturm& s, double t, int& n, int& last) {
    if (I < s.len) {
        const double v = horner_lead<10 - I>(s.g[I], s.deg[I], t);
        const int sg = v > 0.0 ? 1 : (v < 0.0 ? -1 : 0);
        if (sg != 0) {
            if (last != 0 && sg != last) ++n;
            last = sg;
        }
    }
}
POPE_HD int sturm_changes(const Sturm& s, double t) {
    int n = 0, last = 0;
    sturm_sign<0>(s, t, n, last); sturm_sign<1>(s, t, n, last); sturm_sign<2>(s, t, n, last); sturm_sign<3>(s, t, n, last);
    sturm_sign<4>(s, t, n, last); sturm_sign<5>(s, t, n, last); sturm_sign<6>(s, t, n, last); sturm_sign<7>(s, t, n, last);
    sturm_sign<8>(s, t, n, last); sturm_sign<9>(s, t, n, last); sturm_sign<10>(s, t, n, last);
    return n;
}
// value of a polynomial given lowest coefficient first (the small fixed-degree ones of the solver)
POPE_HD double horner(const double* c, int d, double t) {
    double v = c[d];
    for (int k = d - 1; k >= 0; --k) v = v * t + c[k];
    return v;
}
// distinct real roots of c (degree d) in (lo, hi], ascending -> out[]; returns their number (<= 10)
POPE_HD int sturm_roots(const double* c, int d, double lo, double hi, double* out) {
    Sturm s;
    sturm_build(s, c, d);
    if (s.deg[0] == 0) return 0;
    const double* p = s.g[0];
    const int dp = s.deg[0];
    double slo[12], shi[12];
    int vlo[12], vhi[12];
    int sp = 0, n = 0;
    slo[0] = lo; shi[0] = hi; vlo[0] = sturm_changes(s, lo); vhi[0] = sturm_changes(s, hi); sp = 1;
    while (sp > 0 && n < 10) {
        --sp;
        double a = slo[sp], b = shi[sp];
        int va = vlo[sp], vb = vhi[sp];
        if (va - vb <= 0) continue;
        // narrow (a, b] until it holds exactly one root (or is too small to split: a cluster, reported once)
        while (va - vb > 1 && b - a > 1e-13) {
            const double m = 0.5 * (a + b);
            const int vm = sturm_changes(s, m);
            if (va - vm > 0 && vm - vb > 0) {     // roots on both sides: keep the left half, push the right one
                if (sp < 12) { slo[sp] = m; shi[sp] = b; vlo[sp] = vm; vhi[sp] = vb; ++sp; }
                b = m; vb = vm;
            } else if (va - vm > 0) { b = m; vb = vm; }
            else { a = m; va = vm; }
        }
        double fa = horner_lead<10>(p, dp, a), fb = horner_lead<10>(p, dp, b);
        double root;
        if (fb == 0.0) root = b;
        else if ((fa < 0.0) != (fb < 0.0) && fa != 0.0) {
            for (int it = 0; it < 10; ++it) {     // bisection, then Newton inside the bracket
                const double m = 0.5 * (a + b), fm = horner_lead<10>(p, dp, m);
                if ((fm < 0.0) == (fa < 0.0)) { a = m; fa = fm; } else { b = m; fb = fm; }
            }
            root = 0.5 * (a + b);
            for (int it = 0; it < 12; ++it) {
                double v = p[0], dv = 0.0;
                POPE_UNROLL
                for (int m = 1; m <= 10; ++m)
                    if (m <= dp) { dv = dv * root + v; v = v * root + p[m]; }
                if ((v < 0.0) == (fa < 0.0)) { a = root; } else { b = root; }
                double nx = dv != 0.0 ? root - v / dv : 0.5 * (a + b);
                if (!(nx > a && nx < b)) nx = 0.5 * (a + b);
                if (fabs(nx - root) <= 4e-16 * fmax(1.0, fabs(root))) { root = nx; break; }
                root = nx;
            }
        } else {                                   // no sign change (even multiplicity / cluster): Sturm bisection to the end
            while (b - a > 1e-13) {
                const double m = 0.5 * (a + b);
                const int vm = sturm_changes(s, m);
                if (va - vm > 0) { b = m; vb = vm; } else { a = m; va = vm; }
            }
            root = 0.5 * (a + b);
        }
        out[n++] = root;
    }
    for (int i = 1; i < n; ++i) {                  // the stack pops left halves first, but keep the contract explicit
        const double v = out[i];
        int j = i - 1;
        while (j >= 0 && out[j] > v) { out[j + 1] = out[j]; --j; }
        out[j + 1] = v;
    }
    return n;
}

// ---- five-point solver ------------------------------------------------------------------------------------------------
// x0, x1: five correspondences in normalised coordinates ([5][2] each).  E_out[k][9] (row-major, unit Frobenius norm),
// ascending root order; returns k <= 10.  Same algorithm as oracle/pose_ref.py:five_point (null space by Householder
// instead of SVD, roots by Sturm sequences instead of companion eigenvalues — both basis / method independent).
POPE_HD int five_point(const double* x0, const double* x1, double (*E_out)[9]) {
    // M = Q' (9 x 5), column j = the epipolar constraint of correspondence j
    double M[9][5];
    for (int j = 0; j < 5; ++j) {
        const double a = x0[2 * j], b = x0[2 * j + 1], c = x1[2 * j], d = x1[2 * j + 1];
        M[0][j] = c * a; M[1][j] = c * b; M[2][j] = c; M[3][j] = d * a; M[4][j] = d * b; M[5][j] = d; M[6][j] = a; M[7][j] = b; M[8][j] = 1.0;
    }
    double V[5][9];   // Householder vectors
    for (int j = 0; j < 5; ++j) {
        double nrm = 0.0;
        for (int i = j; i < 9; ++i) nrm += M[i][j] * M[i][j];
        nrm = sqrt(nrm);
        if (!(nrm > 1e-300)) return 0;
        const double alpha = M[j][j] > 0.0 ? -nrm : nrm;
        double vv = 0.0;
        for (int i = 0; i < 9; ++i) V[j][i] = i < j ? 0.0 : M[i][j];
        V[j][j] -= alpha;
        for (int i = j; i < 9; ++i) vv += V[j][i] * V[j][i];
        if (!(vv > 1e-300)) return 0;
        const double beta = 2.0 / vv;
        for (int c = j; c < 5; ++c) {
            double dot = 0.0;
            for (int i = j; i < 9; ++i) dot += V[j][i] * M[i][c];
            dot *= beta;
            for (int i = j; i < 9; ++i) M[i][c] -= dot * V[j][i];
        }
        for (int i = j; i < 9; ++i) V[j][i] *= sqrt(beta);   // H = I - v v'
    }
    // basis[b][.] = column 5 + b of H1 H2 .. H5 (orthonormal null space of Q); E = x X + y Y + z Z + W
    double basis[4][9];
    for (int b = 0; b < 4; ++b) {
        for (int i = 0; i < 9; ++i) basis[b][i] = i == 5 + b ? 1.0 : 0.0;
        for (int j = 4; j >= 0; --j) {
            double dot = 0.0;
            for (int i = j; i < 9; ++i) dot += V[j][i] * basis[b][i];
            for (int i = j; i < 9; ++i) basis[b][i] -= dot * V[j][i];
        }
    }
    // entries of E as linear polynomials e[rc][4]
    double e[9][4];
    for (int k = 0; k < 9; ++k)
        for (int b = 0; b < 4; ++b) e[k][b] = basis[b][k];
    double A[10][20];
    for (int r = 0; r < 10; ++r)
        for (int c = 0; c < 20; ++c) A[r][c] = 0.0;
    {   // det(E) = 0
        double q[10];
        for (int i = 0; i < 10; ++i) q[i] = 0.0;
        mac_ll(q, e[0], e[4], 1.0); mac_ll(q, e[1], e[3], -1.0); mac_ql(A[0], q, e[8], 1.0);
        for (int i = 0; i < 10; ++i) q[i] = 0.0;
        mac_ll(q, e[1], e[5], 1.0); mac_ll(q, e[2], e[4], -1.0); mac_ql(A[0], q, e[6], 1.0);
        for (int i = 0; i < 10; ++i) q[i] = 0.0;
        mac_ll(q, e[2], e[3], 1.0); mac_ll(q, e[0], e[5], -1.0); mac_ql(A[0], q, e[7], 1.0);
    }
    {   // 2 E E' E - tr(E E') E = 0  <=>  (E E' - tr/2 I) E = 0
        double lam[9][10];
        for (int r = 0; r < 3; ++r)
            for (int c = r; c < 3; ++c) {
                double* q = lam[3 * r + c];
                for (int i = 0; i < 10; ++i) q[i] = 0.0;
                for (int k = 0; k < 3; ++k) mac_ll(q, e[3 * r + k], e[3 * c + k], 1.0);
            }
        for (int i = 0; i < 10; ++i) {
            const double half_tr = 0.5 * (lam[0][i] + lam[4][i] + lam[8][i]);
            lam[0][i] -= half_tr; lam[4][i] -= half_tr; lam[8][i] -= half_tr;
            lam[3][i] = lam[1][i]; lam[6][i] = lam[2][i]; lam[7][i] = lam[5][i];
        }
        for (int r = 0; r < 3; ++r)
            for (int c = 0; c < 3; ++c)
                for (int k = 0; k < 3; ++k) mac_ql(A[1 + 3 * r + c], lam[3 * r + k], e[3 * k + c], 1.0);
    }
    // Gauss-Jordan on the ten leading monomials (partial pivoting).  Every index is a compile-time constant after unrolling —
    // the pivot row is brought up by predicated swaps instead of A[piv][k] — so the matrix is not forced into scratch memory
    // by a dynamic row index
    POPE_UNROLL
    for (int c = 0; c < 10; ++c) {
        int piv = c;
        double best = fabs(A[c][c]);
        POPE_UNROLL
        for (int r = c + 1; r < 10; ++r)
            if (fabs(A[r][c]) > best) { best = fabs(A[r][c]); piv = r; }
        if (!(best > 1e-300)) return 0;
        POPE_UNROLL
        for (int r = c + 1; r < 10; ++r) {
            const bool sw = piv == r;
            POPE_UNROLL
            for (int k = c; k < 20; ++k) {
                const double u = A[c][k], v = A[r][k];
                A[c][k] = sw ? v : u;
                A[r][k] = sw ? u : v;
            }
        }
        const double inv = 1.0 / A[c][c];
        POPE_UNROLL
        for (int k = c; k < 20; ++k) A[c][k] *= inv;
        POPE_UNROLL
        for (int r = 0; r < 10; ++r) {
            if (r == c) continue;
            const double f = A[r][c];
            POPE_UNROLL
            for (int k = c; k < 20; ++k) A[r][k] -= f * A[c][k];
        }
    }
    // <k> = <e> - z <f>, <l> = <g> - z <h>, <m> = <i> - z <j>:  x bx(z) + y by(z) + b1(z) = 0, coefficients low -> high
    double bx[3][4], by[3][4], b1[3][5];
    for (int r = 0; r < 3; ++r) {
        const double* a = &A[4 + 2 * r][10];
        const double* b = &A[5 + 2 * r][10];
        bx[r][3] = -b[0]; bx[r][2] = a[0] - b[1]; bx[r][1] = a[1] - b[2]; bx[r][0] = a[2];
        by[r][3] = -b[3]; by[r][2] = a[3] - b[4]; by[r][1] = a[4] - b[5]; by[r][0] = a[5];
        b1[r][4] = -b[6]; b1[r][3] = a[6] - b[7]; b1[r][2] = a[7] - b[8]; b1[r][1] = a[8] - b[9]; b1[r][0] = a[9];
    }
    // det B(z): sum over the cofactor expansion along the last column
    double poly[11];
    for (int k = 0; k <= 10; ++k) poly[k] = 0.0;
    for (int r = 0; r < 3; ++r) {
        const int r1 = (r + 1) % 3, r2 = (r + 2) % 3;
        double m[7];   // bx[r1] by[r2] - bx[r2] by[r1]   (cyclic order keeps the cofactor sign +)
        for (int k = 0; k < 7; ++k) m[k] = 0.0;
        for (int i = 0; i < 4; ++i)
            for (int j = 0; j < 4; ++j) m[i + j] += bx[r1][i] * by[r2][j] - bx[r2][i] * by[r1][j];
        for (int i = 0; i < 7; ++i)
            for (int j = 0; j < 5; ++j) poly[i + j] += m[i] * b1[r][j];
    }
    for (int k = 0; k <= 10; ++k)
        if (!(fabs(poly[k]) < 1e300)) return 0;     // NaN / inf from a degenerate sample
    // real roots: |z| <= 1 from p(z) on (-1, 1]; |z| > 1 from the reversed polynomial in u = 1 / z on (-1, 1)
    double zs[20];
    int nz = 0;
    {
        double rr[2][10];
        int nr[2];
        POPE_ROLLED
        for (int pass = 0; pass < 2; ++pass) {      // one copy of the root finder in the code: u first, then z
            double cur[11];
            for (int k = 0; k <= 10; ++k) cur[k] = pass ? poly[k] : poly[10 - k];
            nr[pass] = sturm_roots(cur, 10, -1.0, 1.0, rr[pass]);
        }
        const double* ru = rr[0];
        const double* rz = rr[1];
        const int nu = nr[0], n1 = nr[1];
        // ascending z: z < -1 are the u in (-1, 0) taken from the one closest to 0 backwards, then (-1, 1], then z > 1 = the u
        // in (0, 1) again from the largest downwards (u = 1 is z = 1, already counted; u = 0 is a root at infinity)
        for (int i = nu - 1; i >= 0; --i)
            if (ru[i] < 0.0 && ru[i] > -1.0) zs[nz++] = 1.0 / ru[i];
        for (int i = 0; i < n1; ++i) zs[nz++] = rz[i];
        for (int i = nu - 1; i >= 0; --i)
            if (ru[i] > 0.0 && ru[i] < 1.0) zs[nz++] = 1.0 / ru[i];
    }
    int n = 0;
    for (int i = 0; i < nz && n < 10; ++i) {
        const double z = zs[i];
        double Bz[3][3];
        for (int r = 0; r < 3; ++r) {
            Bz[r][0] = horner(bx[r], 3, z); Bz[r][1] = horner(by[r], 3, z); Bz[r][2] = horner(b1[r], 4, z);
        }
        // null vector of the (rank 2) matrix: the largest cross product of two rows
        double nv[3] = {0, 0, 0}, nbest = -1.0;
        for (int r = 0; r < 3; ++r) {
            const double* a = Bz[r];
            const double* b = Bz[(r + 1) % 3];
            const double c0 = a[1] * b[2] - a[2] * b[1], c1 = a[2] * b[0] - a[0] * b[2], c2 = a[0] * b[1] - a[1] * b[0];
            const double nn = c0 * c0 + c1 * c1 + c2 * c2;
            if (nn > nbest) { nbest = nn; nv[0] = c0; nv[1] = c1; nv[2] = c2; }
        }
        const double big = fmax(fabs(nv[0]), fmax(fabs(nv[1]), fabs(nv[2])));
        if (!(fabs(nv[2]) >= 1e-12 * big) || !(big > 0.0)) continue;
        double nrm = 0.0;
        double* E = E_out[n];
        for (int k = 0; k < 9; ++k) {
            E[k] = nv[0] * basis[0][k] + nv[1] * basis[1][k] + z * nv[2] * basis[2][k] + nv[2] * basis[3][k];
            nrm += E[k] * E[k];
        }
        nrm = sqrt(nrm);
        if (!(nrm > 0.0) || !(nrm < 1e300)) continue;
        for (int k = 0; k < 9; ++k) E[k] /= nrm;
        ++n;
    }
    return n;
}

// ---- scoring --------------------------------------------------------------------------------------------------------------
// Sampson error of one correspondence: (x1h' E x0h)^2 / (|E x0h|_xy^2 + |E' x1h|_xy^2)
POPE_HD double sampson(const double* E, double ax, double ay, double bx, double by) {
    const double e0 = E[0] * ax + E[1] * ay + E[2], e1 = E[3] * ax + E[4] * ay + E[5], e2 = E[6] * ax + E[7] * ay + E[8];
    const double t0 = E[0] * bx + E[3] * by + E[6], t1 = E[1] * bx + E[4] * by + E[7];
    const double num = bx * e0 + by * e1 + e2;
    return num * num / (e0 * e0 + e1 * e1 + t0 * t0 + t1 * t1);
}

// the inlier test error <= t2 without the fp64 division (den > 0: a degenerate correspondence is never an inlier)
POPE_HD bool sampson_inlier(const double* E, double ax, double ay, double bx, double by, double t2) {
    // fused multiply-adds spelled out (the library is built with -ffp-contract=off): 18 fp64 instructions per correspondence,
    // the same on the device and in the host build
    const double e0 = fma(E[0], ax, fma(E[1], ay, E[2])), e1 = fma(E[3], ax, fma(E[4], ay, E[5])), e2 = fma(E[6], ax, fma(E[7], ay, E[8]));
    const double t0 = fma(E[0], bx, fma(E[3], by, E[6])), t1 = fma(E[1], bx, fma(E[4], by, E[7]));
    const double num = fma(bx, e0, fma(by, e1, e2)), den = fma(e0, e0, fma(e1, e1, fma(t0, t0, t1 * t1)));
    return den > 0.0 && num * num <= t2 * den;
}

// RANSACUpdateNumIters with five model points
POPE_HD int update_num_iters(double conf, double outlier_ratio, int max_iters) {
    const double p = fmin(fmax(conf, 0.0), 1.0), ep = fmin(fmax(outlier_ratio, 0.0), 1.0);
    double num = fmax(1.0 - p, 2.2250738585072014e-308);
    const double w = 1.0 - ep;
    double denom = 1.0 - w * w * w * w * w;
    if (denom < 2.2250738585072014e-308) return 0;
    num = log(num);
    denom = log(denom);
    if (denom >= 0.0 || -num >= max_iters * (-denom)) return max_iters;
    return int(rint(num / denom));
}

// ---- decomposition and triangulation ----------------------------------------------------------------------------------------
// one-sided Jacobi on an n x n matrix stored a[row][col] (n <= 4): on return the columns of `a` are orthogonal (U S) and `v`
// holds the right singular vectors as columns
template <int N>
POPE_HD void jacobi_svd(double (*a)[N], double (*v)[N]) {
    POPE_UNROLL
    for (int i = 0; i < N; ++i) {
        POPE_UNROLL
        for (int j = 0; j < N; ++j) v[i][j] = i == j ? 1.0 : 0.0;
    }
    for (int sweep = 0; sweep < 30; ++sweep) {
        bool rotated = false;
        POPE_UNROLL
        for (int p = 0; p < N - 1; ++p) {
            POPE_UNROLL
            for (int q = p + 1; q < N; ++q) {
                double alpha = 0.0, beta = 0.0, gamma = 0.0;
                POPE_UNROLL
                for (int i = 0; i < N; ++i) { alpha += a[i][p] * a[i][p]; beta += a[i][q] * a[i][q]; gamma += a[i][p] * a[i][q]; }
                if (!(fabs(gamma) > 1e-17 * sqrt(alpha * beta)) || gamma == 0.0) continue;
                rotated = true;
                const double zeta = (beta - alpha) / (2.0 * gamma);
                const double t = (zeta >= 0.0 ? 1.0 : -1.0) / (fabs(zeta) + sqrt(1.0 + zeta * zeta));
                const double c = 1.0 / sqrt(1.0 + t * t), s = c * t;
                POPE_UNROLL
                for (int i = 0; i < N; ++i) {
                    const double ap = a[i][p], aq = a[i][q];
                    a[i][p] = c * ap - s * aq; a[i][q] = s * ap + c * aq;
                    const double vp = v[i][p], vq = v[i][q];
                    v[i][p] = c * vp - s * vq; v[i][q] = s * vp + c * vq;
                }
            }
        }
        if (!rotated) break;
    }
}

// decomposeEssentialMat: R1 = U W V', R2 = U W' V', t = u3, with det(U) = det(V) = +1 (row-major 3x3 outputs)
POPE_HD void decompose_essential(const double* E, double* R1, double* R2, double* t) {
    double a[3][3], v[3][3];
    for (int i = 0; i < 3; ++i)
        for (int j = 0; j < 3; ++j) a[i][j] = E[3 * i + j];
    jacobi_svd<3>(a, v);
    double s[3];
    int ord[3] = {0, 1, 2};
    for (int j = 0; j < 3; ++j) s[j] = sqrt(a[0][j] * a[0][j] + a[1][j] * a[1][j] + a[2][j] * a[2][j]);
    for (int i = 0; i < 2; ++i)
        for (int j = i + 1; j < 3; ++j)
            if (s[ord[j]] > s[ord[i]]) { const int k = ord[i]; ord[i] = ord[j]; ord[j] = k; }
    double u1[3], u2[3], u3[3], v1[3], v2[3], v3[3];
    for (int i = 0; i < 3; ++i) {
        u1[i] = a[i][ord[0]] / s[ord[0]]; u2[i] = a[i][ord[1]] / s[ord[1]];
        v1[i] = v[i][ord[0]]; v2[i] = v[i][ord[1]];
    }
    // re-orthogonalise u2 against u1 (the two leading singular values of an essential matrix are equal: any rotation of the
    // pair is a valid choice, but they must be orthonormal)
    double d = u1[0] * u2[0] + u1[1] * u2[1] + u1[2] * u2[2];
    for (int i = 0; i < 3; ++i) u2[i] -= d * u1[i];
    d = sqrt(u2[0] * u2[0] + u2[1] * u2[1] + u2[2] * u2[2]);
    for (int i = 0; i < 3; ++i) u2[i] /= d;
    u3[0] = u1[1] * u2[2] - u1[2] * u2[1]; u3[1] = u1[2] * u2[0] - u1[0] * u2[2]; u3[2] = u1[0] * u2[1] - u1[1] * u2[0];
    v3[0] = v1[1] * v2[2] - v1[2] * v2[1]; v3[1] = v1[2] * v2[0] - v1[0] * v2[2]; v3[2] = v1[0] * v2[1] - v1[1] * v2[0];
    for (int i = 0; i < 3; ++i)
        for (int j = 0; j < 3; ++j) {
            const double w = u1[i] * v2[j] - u2[i] * v1[j], z = u3[i] * v3[j];
            R1[3 * i + j] = w + z;     // U W V'  = -u2 v1' + u1 v2' + u3 v3'
            R2[3 * i + j] = -w + z;    // U W' V' =  u2 v1' - u1 v2' + u3 v3'
        }
    t[0] = u3[0]; t[1] = u3[1]; t[2] = u3[2];
}

// cv::triangulatePoints for one correspondence with P0 = [I | 0], P1 = [R | t]: the right singular vector of the smallest
// singular value of the 4 x 4 DLT system A, then recoverPose's tests — in front of both cameras and closer than `dist`.
// The vector is found by inverse iteration on A'A through ONE LU factorisation of A (P A = L U  =>  (A'A)^-1 =
// U^-1 L^-1 L^-T U^-T: the row permutation cancels): the smallest singular value of a triangulation system lies orders of
// magnitude below the others for an inlier, so two iterations reach the vector an SVD returns (eight are run: clutter
// converges slower, and it is never counted anyway) (a vanishing pivot — exact,
// noise-free data — is replaced by a tiny one, the classical remedy).  ~250 fp64 operations with 4 divisions instead of a
// 4 x 4 Jacobi SVD's ~9 000 with ~150 square roots and divisions: this was the longest phase of the kernel.
POPE_HD bool cheirality(const double* R, const double* t, double ax, double ay, double bx, double by, double dist) {
    double a[4][4] = {{-1.0, 0.0, ax, 0.0}, {0.0, -1.0, ay, 0.0},
                      {bx * R[6] - R[0], bx * R[7] - R[1], bx * R[8] - R[2], bx * t[2] - t[0]},
                      {by * R[6] - R[3], by * R[7] - R[4], by * R[8] - R[5], by * t[2] - t[1]}};
    double amax = 0.0;
    POPE_UNROLL
    for (int i = 0; i < 4; ++i) {
        POPE_UNROLL
        for (int j = 0; j < 4; ++j) amax = fmax(amax, fabs(a[i][j]));
    }
    if (!(amax < 1e300)) return false;               // NaN / inf coordinates
    const double tiny = 1e-30 * amax + 1e-300;
    double inv[4];
    POPE_UNROLL
    for (int k = 0; k < 4; ++k) {
        int piv = k;
        double best = fabs(a[k][k]);
        POPE_UNROLL
        for (int r = k + 1; r < 4; ++r)
            if (fabs(a[r][k]) > best) { best = fabs(a[r][k]); piv = r; }
        POPE_UNROLL
        for (int r = k + 1; r < 4; ++r)
            if (piv == r) {
                POPE_UNROLL
                for (int c = 0; c < 4; ++c) { const double s = a[k][c]; a[k][c] = a[r][c]; a[r][c] = s; }
            }
        double d = a[k][k];
        if (!(fabs(d) >= tiny)) d = d < 0.0 ? -tiny : tiny;
        a[k][k] = d;
        inv[k] = 1.0 / d;
        POPE_UNROLL
        for (int r = k + 1; r < 4; ++r) {
            const double f = a[r][k] * inv[k];
            a[r][k] = f;
            POPE_UNROLL
            for (int c = k + 1; c < 4; ++c) a[r][c] -= f * a[k][c];
        }
    }
    double x[4] = {1.0, 1.0, 1.0, 1.0};
    for (int it = 0; it < 8; ++it) {   // inliers (the only points recoverPose counts) converge in two; clutter gets the rest
        // z = U^-T x
        x[0] *= inv[0];
        x[1] = (x[1] - a[0][1] * x[0]) * inv[1];
        x[2] = (x[2] - a[0][2] * x[0] - a[1][2] * x[1]) * inv[2];
        x[3] = (x[3] - a[0][3] * x[0] - a[1][3] * x[1] - a[2][3] * x[2]) * inv[3];
        // w = L^-T z (unit upper)
        x[2] -= a[3][2] * x[3];
        x[1] -= a[2][1] * x[2] + a[3][1] * x[3];
        x[0] -= a[1][0] * x[1] + a[2][0] * x[2] + a[3][0] * x[3];
        // v = L^-1 w (unit lower)
        x[1] -= a[1][0] * x[0];
        x[2] -= a[2][0] * x[0] + a[2][1] * x[1];
        x[3] -= a[3][0] * x[0] + a[3][1] * x[1] + a[3][2] * x[2];
        // x = U^-1 v
        x[3] *= inv[3];
        x[2] = (x[2] - a[2][3] * x[3]) * inv[2];
        x[1] = (x[1] - a[1][2] * x[2] - a[1][3] * x[3]) * inv[1];
        x[0] = (x[0] - a[0][1] * x[1] - a[0][2] * x[2] - a[0][3] * x[3]) * inv[0];
        const double m = fmax(fmax(fabs(x[0]), fabs(x[1])), fmax(fabs(x[2]), fabs(x[3])));
        if (!(m > 0.0) || !(m < 1e300)) return false;
        const double s = 1.0 / m;
        x[0] *= s; x[1] *= s; x[2] *= s; x[3] *= s;
    }
    const double X = x[0], Y = x[1], Z = x[2], W = x[3];
    if (!(Z * W > 0.0)) return false;
    const double px = X / W, py = Y / W, pz = Z / W;
    if (!(pz < dist)) return false;
    const double z1 = R[6] * px + R[7] * py + R[8] * pz + t[2];
    return z1 > 0.0 && z1 < dist;
}

}  // namespace pose
