// Batched relative-pose solver: `estimate_pose` of the reference (src/utils/metrics.py:69-94 — K
// normalisation, cv2.findEssentialMat RANSAC, cv2.recoverPose over the returned E's) for B pairs in ONE launch, fed straight
// from the matcher's compacted (counts, mkpts0, mkpts1) buffers: no host round trip between matching and pose.
//
// One 256-thread workgroup per pair.  A RANSAC round = 256 hypotheses, one per thread: five correspondences chosen by a
// counter-based hash (batch-invariant, reproducible), the Nister five-point solver in fp64 (pose_math.h: Householder null
// space, ten cubic constraints, Gauss-Jordan, 10th-degree polynomial, Sturm-sequence roots) -> up to ten essential matrices
// kept in the thread's private memory; every thread then scores its own candidates against all N correspondences, which
// stream through LDS in tiles of 256 (32 bytes each, read as wave-wide broadcasts).  The best (inliers, lowest hypothesis,
// lowest root) wins through one 64-bit LDS max; the iteration budget follows OpenCV's RANSACUpdateNumIters and is tested
// after every round.  recoverPose = decomposition by one thread, linear triangulation + cheirality of every correspondence
// under the four (R, t) combinations by all threads, counts through LDS atomics.  Everything after the fp32 inputs is fp64:
// this step is latency-bound integer / divide / root-finding work, not throughput — at 1 400 pairs/s it needs < 0.1 % of the
// chip's fp64 rate, and a pair's whole solve is a few hundred microseconds of one CU.
#include "common.h"
#include "kernels.h"
#include "pose_math.h"

namespace {

constexpr int NT = pose::ROUND;

#ifdef POSE_STAMPS   // dev: wall-clock stamps (10 ns ticks) of workgroup 0's phases, returned in E[0] of pair 0 (scripts/pose_time.py)
#define POSE_STAMP(k) do { if (blockIdx.x == 0 && threadIdx.x == 0) stamps[k] = wall_clock64(); } while (0)
#else
#define POSE_STAMP(k) do {} while (0)
#endif

#define POSE_UNROLL _Pragma("unroll")

struct __attribute__((aligned(32))) Pt { double ax, ay, bx, by; };

// read of the per-pair model list written by other waves of the workgroup (device scope: not served from a stale L1 line)
__device__ inline double list_load(const double* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }

__global__ __launch_bounds__(NT) void pose_kernel(PoseParams q) {
    __shared__ Pt s_pts[NT + 4];           // + 4: the scoring loop prefetches past the tile's end (values unused)
    __shared__ double s_E[10][9];          // candidates handed to recoverPose (1 after RANSAC, <= 10 for the minimal problem)
    __shared__ double s_R1[9], s_R2[9], s_t[3];
    __shared__ unsigned long long s_key;
    __shared__ int s_off, s_ncand, s_good[4], s_total, s_badprefix;
    const int b = blockIdx.x, tid = threadIdx.x;
#ifdef POSE_STAMPS
    __shared__ unsigned long long stamps[8];
#endif
    POSE_STAMP(0);
    if (tid == 0) { s_off = 0; s_key = 0ull; s_ncand = 0; s_badprefix = 0; }
    __syncthreads();
    {
        // offset of this pair's matches = sum of the counts before it; a NEGATIVE count anywhere before it would pull the
        // offset back under an earlier pair's rows (overlapping or negative: out-of-bounds writes), so it refuses this pair too
        int part = 0, bad = 0;
        for (int k = tid; k < b; k += NT) {
            const int c = q.counts[k];
            bad |= c < 0;
            part += c > 0 ? c : 0;
        }
        if (part) atomicAdd(&s_off, part);
        if (bad) s_badprefix = 1;
    }
    __syncthreads();
    const long long off = s_off;
    const int N = q.counts[b];
    int* info = q.info + 8 * b;
    double* Rout = q.R + 9 * b;
    double* tout = q.t + 3 * b;
    double* Eout = q.E + 9 * b;
    if (tid == 0) {
        for (int k = 0; k < 8; ++k) info[k] = 0;
        info[6] = N;
        for (int k = 0; k < 9; ++k) { Rout[k] = 0.0; Eout[k] = 0.0; }
        tout[0] = tout[1] = tout[2] = 0.0;
    }
    if (N < 0 || s_badprefix || off < 0 || off + N > q.M) {   // counts the caller's capacity does not cover, or corrupt: refuse the pair
        if (tid == 0) info[7] = -1;
        return;
    }
    Pt* xn = reinterpret_cast<Pt*>(q.xn) + off;
    unsigned char* mask = q.mask_ws + off;
    unsigned char* cheir = q.cheir_ws + off;
    unsigned char* inl = q.inliers + off;
    double* clist = q.cand_ws + size_t(b) * (NT * 10 * 9);   // this round's models, [<= NT * 10][9]
    int* cmeta = q.cmeta_ws + size_t(b) * (NT * 10);          // their (hypothesis * 16 + root)
    const double* K0 = q.K0 + 9 * b;
    const double* K1 = q.K1 + 9 * b;
    // metrics.py:72-75: (kpts - [cx, cy]) / [fx, fy] in fp64; :78 threshold / mean(fx0, fy1, fx0, fy1)
    for (int i = tid; i < N; i += NT) {
        Pt p;
        p.ax = (double(q.kpts0[2 * (off + i)]) - K0[2]) / K0[0];
        p.ay = (double(q.kpts0[2 * (off + i) + 1]) - K0[5]) / K0[4];
        p.bx = (double(q.kpts1[2 * (off + i)]) - K1[2]) / K1[0];
        p.by = (double(q.kpts1[2 * (off + i) + 1]) - K1[5]) / K1[4];
        xn[i] = p;
        inl[i] = 0;
        mask[i] = 1;
    }
    if (N < 5) return;                       // metrics.py:70-71 -> None
    const double thr = q.thresh / ((K0[0] + K1[4] + K0[0] + K1[4]) / 4.0);
    const double t2 = thr * thr;
    __syncthreads();

    const bool minimal = N == 5;             // the minimal problem itself: every solution goes to recoverPose, all five
                                             // points count as inliers (one call site of the solver serves both cases)
    {
        int niters = q.max_iters, done = 0, rounds = 0;
        unsigned long long best_key = 0ull;
        double cand[10][9];
        while (done < niters) {
            if (tid == 0) s_total = 0;
            __syncthreads();
            const int h = done + tid;
            int ncand = 0;
            if (minimal ? tid == 0 : h < q.max_iters) {
                int pick[5] = {0, 1, 2, 3, 4};
                if (!minimal) pose::sample_indices(q.seed, unsigned(h), unsigned(N), pick);
                double x0[10], x1[10];
                for (int i = 0; i < 5; ++i) {
                    const Pt p = xn[pick[i]];
                    x0[2 * i] = p.ax; x0[2 * i + 1] = p.ay; x1[2 * i] = p.bx; x1[2 * i + 1] = p.by;
                }
                ncand = pose::five_point(x0, x1, cand);
            }
            if (done == 0) POSE_STAMP(1);
            if (minimal) {
                if (tid == 0) {
                    for (int k = 0; k < ncand; ++k)
                        for (int j = 0; j < 9; ++j) s_E[k][j] = cand[k][j];
                    s_ncand = ncand;
                    info[1] = 5; info[2] = 1; info[3] = 0; info[4] = 0; info[5] = 0;
                }
                break;
            }
            // Score the round's models with the work spread evenly: a thread holds 0..10 roots (about 4 on average, and a
            // wave would wait for its slowest lane), so the models go through a per-pair list in the workspace and thread i
            // scores entries i, i + NT, ... of it -- ceil(total / NT) each, two at a time per point read.
            int slot = 0;
            if (ncand) slot = atomicAdd(&s_total, ncand);
            for (int k = 0; k < ncand; ++k) {
                for (int j = 0; j < 9; ++j) clist[size_t(slot + k) * 9 + j] = cand[k][j];
                cmeta[slot + k] = h * 16 + k;
            }
            __threadfence();                 // the list is read by other waves: written back before the barrier, and read
            __syncthreads();                 // below with agent-scope loads (the same addresses may sit stale in L1 from the
            const int total = s_total;       // round before)
            const int per = (total + NT - 1) / NT;
            int cnt[10];
            POSE_UNROLL
            for (int k = 0; k < 10; ++k) cnt[k] = 0;
            for (int base = 0; base < N; base += NT) {
                __syncthreads();
                if (base + tid < N) s_pts[tid] = xn[base + tid];
                __syncthreads();
                const int m = min(NT, N - base);
                POSE_UNROLL
                for (int r = 0; r < 10; r += 2) {
                    if (r < per) {           // uniform over the workgroup
                        const int ca = tid + r * NT, cb = ca + NT;
                        double Ea[9], Eb[9];
                        POSE_UNROLL
                        for (int j = 0; j < 9; ++j) {
                            Ea[j] = ca < total ? list_load(clist + size_t(ca) * 9 + j) : 0.0;   // a zero model has no inliers
                            Eb[j] = cb < total ? list_load(clist + size_t(cb) * 9 + j) : 0.0;
                        }
                        int na = 0, nb = 0;
                        // one wave per SIMD: nothing else hides the LDS latency, so the next two points are fetched while
                        // these two are scored (s_pts is padded by two entries for the reads past the end)
                        Pt p0 = s_pts[0], p1 = s_pts[1];
                        for (int i = 0; i < m; i += 2) {
                            const Pt c0 = p0, c1 = p1;
                            p0 = s_pts[i + 2]; p1 = s_pts[i + 3];
                            na += pose::sampson_inlier(Ea, c0.ax, c0.ay, c0.bx, c0.by, t2) ? 1 : 0;
                            nb += pose::sampson_inlier(Eb, c0.ax, c0.ay, c0.bx, c0.by, t2) ? 1 : 0;
                            if (i + 1 < m) {
                                na += pose::sampson_inlier(Ea, c1.ax, c1.ay, c1.bx, c1.by, t2) ? 1 : 0;
                                nb += pose::sampson_inlier(Eb, c1.ax, c1.ay, c1.bx, c1.by, t2) ? 1 : 0;
                            }
                        }
                        cnt[r] += na; cnt[r + 1] += nb;
                    }
                }
            }
            if (done == 0) POSE_STAMP(2);
            // the first model with the most inliers: (count, lowest hypothesis, lowest root) as one comparable key
            unsigned long long key = 0ull;
            int kc = -1;
            POSE_UNROLL
            for (int r = 0; r < 10; ++r) {
                const int c = tid + r * NT;
                if (c < total && cnt[r] >= 5) {
                    const unsigned meta = unsigned(__hip_atomic_load(cmeta + c, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
                    const unsigned long long k2 = ((unsigned long long)cnt[r] << 32) | (0xFFFFFFFFull - (unsigned long long)meta);
                    if (k2 > key) { key = k2; kc = c; }
                }
            }
            if (key) atomicMax(&s_key, key);
            __syncthreads();
            const unsigned long long win = s_key;
            if (win > best_key) {
                best_key = win;
                if (key == win) {
                    for (int j = 0; j < 9; ++j) s_E[0][j] = list_load(clist + size_t(kc) * 9 + j);
                    const unsigned meta = 0xFFFFFFFFu - unsigned(win & 0xFFFFFFFFull);
                    info[4] = int(meta >> 4); info[5] = int(meta & 15u);
                }
            }
            done = min(done + NT, q.max_iters);
            ++rounds;
            if (best_key) {
                const int bcount = int(best_key >> 32);
                niters = min(niters, pose::update_num_iters(q.conf, double(N - bcount) / double(N), q.max_iters));
            }
            __syncthreads();
        }
        if (!minimal) {
            if (tid == 0) { info[2] = done; info[3] = rounds; info[1] = int(best_key >> 32); s_ncand = best_key ? 1 : 0; }
            __syncthreads();
            if (!best_key) return;           // "E is None" (metrics.py:82-84)
            double E[9];
            for (int j = 0; j < 9; ++j) E[j] = s_E[0][j];
            for (int i = tid; i < N; i += NT) {
                const Pt p = xn[i];
                mask[i] = pose::sampson_inlier(E, p.ax, p.ay, p.bx, p.by, t2) ? 1 : 0;
            }
        }
        __syncthreads();
    }

    POSE_STAMP(3);
    // metrics.py:86-94: recoverPose for every returned E; the mask is narrowed in place from one E to the next
    const int ncand = s_ncand;
    int best = 0;
    for (int k = 0; k < ncand; ++k) {
        if (tid == 0) {
            pose::decompose_essential(s_E[k], s_R1, s_R2, s_t);
            s_good[0] = s_good[1] = s_good[2] = s_good[3] = 0;
        }
        __syncthreads();
        for (int i = tid; i < N; i += NT) {
            const Pt p = xn[i];
            unsigned bits = 0;
            if (mask[i]) {
                for (int c = 0; c < 4; ++c) {
                    const double* R = (c & 1) ? s_R2 : s_R1;
                    const double sg = (c & 2) ? -1.0 : 1.0;
                    const double t[3] = {sg * s_t[0], sg * s_t[1], sg * s_t[2]};
                    if (pose::cheirality(R, t, p.ax, p.ay, p.bx, p.by, 1e9)) bits |= 1u << c;
                }
            }
            cheir[i] = (unsigned char)bits;
            for (int c = 0; c < 4; ++c)
                if ((bits >> c) & 1u) atomicAdd(&s_good[c], 1);
        }
        __syncthreads();
        POSE_STAMP(4 + (k ? 1 : 0));
        const int g0 = s_good[0], g1 = s_good[1], g2 = s_good[2], g3 = s_good[3];
        const int ch = (g0 >= g1 && g0 >= g2 && g0 >= g3) ? 0 : (g1 >= g0 && g1 >= g2 && g1 >= g3) ? 1 : (g2 >= g0 && g2 >= g1 && g2 >= g3) ? 2 : 3;
        const int n = ch == 0 ? g0 : ch == 1 ? g1 : ch == 2 ? g2 : g3;
        const bool better = n > best;
        for (int i = tid; i < N; i += NT) {
            const unsigned char nm = (cheir[i] >> ch) & 1;
            mask[i] = nm;
            if (better) inl[i] = nm;
        }
        if (better) {
            best = n;
            if (tid == 0) {
                const double* R = (ch & 1) ? s_R2 : s_R1;
                const double sg = (ch & 2) ? -1.0 : 1.0;
                for (int j = 0; j < 9; ++j) { Rout[j] = R[j]; Eout[j] = s_E[k][j]; }
                for (int j = 0; j < 3; ++j) tout[j] = sg * s_t[j];
                info[0] = n;
            }
        }
        __syncthreads();
    }
#ifdef POSE_STAMPS
    POSE_STAMP(6);
    if (blockIdx.x == 0 && tid == 0)
        for (int j = 0; j < 7; ++j) Eout[j] = double(stamps[j] - stamps[0]) * 0.01;   // microseconds
#endif
}

// op-level: one minimal problem per thread (parity tests of the solver itself)
__global__ __launch_bounds__(64) void five_point_kernel(const double* __restrict__ x0, const double* __restrict__ x1, int S,
                                                        double* __restrict__ E_out, int* __restrict__ n_out) {
    const int s = blockIdx.x * 64 + threadIdx.x;
    if (s >= S) return;
    double a[10], b[10], E[10][9];
    for (int i = 0; i < 10; ++i) { a[i] = x0[10 * s + i]; b[i] = x1[10 * s + i]; }
    const int n = pose::five_point(a, b, E);
    n_out[s] = n;
    for (int k = 0; k < 10; ++k)
        for (int j = 0; j < 9; ++j) E_out[(size_t(s) * 10 + k) * 9 + j] = k < n ? E[k][j] : 0.0;
}

inline size_t align256(size_t v) { return (v + 255) & ~size_t(255); }

}  // namespace

size_t pope_pose_workspace(int B, long long M) {
    const size_t m = size_t(M < 1 ? 1 : M), nb = size_t(B < 1 ? 1 : B);
    return align256(m * sizeof(Pt)) + 2 * align256(m) + align256(nb * NT * 10 * 9 * sizeof(double)) + align256(nb * NT * 10 * sizeof(int));
}

int pope_launch_estimate_pose(PoseParams q, void* ws, size_t ws_bytes, hipStream_t stream) {
    if (!q.kpts0 || !q.kpts1 || !q.counts || !q.K0 || !q.K1 || !q.R || !q.t || !q.E || !q.inliers || !q.info || !ws) return POPE_ERR_ARG;
    if (q.B <= 0 || q.M < 0 || q.max_iters < 1 || q.max_iters > (1 << 27) || !(q.thresh > 0.0) || (reinterpret_cast<uintptr_t>(ws) & 31))
        return POPE_ERR_ARG;                 // the winner's key carries hypothesis * 16 + root in 32 bits
    if (ws_bytes < pope_pose_workspace(q.B, q.M)) return POPE_ERR_WORKSPACE;
    const size_t m = size_t(q.M < 1 ? 1 : q.M);
    char* p = static_cast<char*>(ws);
    q.xn = p; p += align256(m * sizeof(Pt));
    q.mask_ws = reinterpret_cast<unsigned char*>(p); p += align256(m);
    q.cheir_ws = reinterpret_cast<unsigned char*>(p); p += align256(m);
    q.cand_ws = reinterpret_cast<double*>(p); p += align256(size_t(q.B) * NT * 10 * 9 * sizeof(double));
    q.cmeta_ws = reinterpret_cast<int*>(p);
    hipLaunchKernelGGL(pose_kernel, dim3(q.B), dim3(NT), 0, stream, q);
    return pope_check_launch();
}

int pope_launch_five_point(const double* x0, const double* x1, int S, double* E_out, int* n_out, hipStream_t stream) {
    if (!x0 || !x1 || !E_out || !n_out || S <= 0) return POPE_ERR_ARG;
    hipLaunchKernelGGL(five_point_kernel, dim3((S + 63) / 64), dim3(64), 0, stream, x0, x1, S, E_out, n_out);
    return pope_check_launch();
}
