// Dev lab (not shipped): where does sam_attn_relpos_kernel's time go?  The shipped kernel's body with switches:
//   V & 1: skip the q loads (zeros)     V & 2: skip the stores (a never-true guard keeps the MFMAs alive)
//   V & 4: one axis only (half the tasks)   V & 8: dense layout (q rows of HD halves, relpos rows of JT halves in a second array)
// hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off scripts/sam_relpos_lab.hip -o scripts/_lab/relpos_lab
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef _Float16 f16x2 __attribute__((ext_vector_type(2)));
typedef _Float16 f16x4 __attribute__((ext_vector_type(4)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
struct Geom { int B, ws, nw, heads, Npad, DQ; };
__device__ __forceinline__ f16x8 cat(f16x4 a, f16x4 b) { return f16x8{a[0], a[1], a[2], a[3], b[0], b[1], b[2], b[3]}; }

template <int HD, int MB, int V>
__global__ __launch_bounds__(256) void relpos(const float* __restrict__ Rh, const float* __restrict__ Rw, _Float16* __restrict__ Qp,
                                              _Float16* __restrict__ Rp, Geom a, int hpg, int n_tasks, int never) {
    constexpr int KS = HD / 16;
    const int lane = threadIdx.x & 63, c = lane & 31, h = lane >> 5;
    int task = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (task >= n_tasks) return;
    const int n_hg = a.heads / hpg;
    const int hg = task % n_hg; task /= n_hg;
    const int r = task % a.ws; task /= a.ws;
    int axis, wb;
    if (V & 4) { axis = 0; wb = task; } else { axis = task & 1; wb = task >> 1; }
    const int JT = a.DQ - HD;
    const int q_row = (V & 8) ? HD : a.DQ;
    f16x8 rh[MB][KS], rl[MB][KS];
    const float* tab = (axis ? Rw : Rh) + (size_t)r * a.ws * HD + 8 * h;
    for (int mb = 0; mb < MB; ++mb) {
        const int j = 32 * mb + c;
#pragma unroll
        for (int s = 0; s < KS; ++s) {
            f32x4 v0 = {0.f, 0.f, 0.f, 0.f}, v1 = v0;
            if (j < a.ws) { v0 = *(const f32x4*)(tab + (size_t)j * HD + 16 * s); v1 = *(const f32x4*)(tab + (size_t)j * HD + 16 * s + 4); }
            f16x4 h0 = __builtin_convertvector(v0 * 256.f, f16x4), h1 = __builtin_convertvector(v1 * 256.f, f16x4);
            f16x4 l0 = __builtin_convertvector(v0 * 256.f - __builtin_convertvector(h0, f32x4), f16x4);
            f16x4 l1 = __builtin_convertvector(v1 * 256.f - __builtin_convertvector(h1, f32x4), f16x4);
            rh[mb][s] = cat(h0, h1); rl[mb][s] = cat(l0, l1);
        }
    }
    const int n_rho = a.ws * hpg;
    const bool quad = !(a.ws & 3);
    for (int rho0 = 0; rho0 < n_rho; rho0 += 32) {
        const int rho = rho0 + c;
        const bool live = rho < n_rho;
        const int hl = rho / a.ws, i = rho - hl * a.ws;
        const int n = axis ? i * a.ws + r : r * a.ws + i;
        const size_t grow = (size_t)(wb * a.heads + hg * hpg + hl) * a.Npad + n;
        _Float16* row = Qp + grow * q_row;
        f16x8 qh[KS];
#pragma unroll
        for (int s = 0; s < KS; ++s) {
            qh[s] = f16x8{0, 0, 0, 0, 0, 0, 0, 0};
            if (!(V & 1) && live) qh[s] = *(const f16x8*)(row + 16 * s + 8 * h);
            if (V & 1) qh[s][0] = _Float16(float(rho & 7));
        }
#pragma unroll
        for (int mb = 0; mb < MB; ++mb) {
            f32x16 acc;
            for (int e = 0; e < 16; ++e) acc[e] = 0.f;
#pragma unroll
            for (int s = 0; s < KS; ++s) { acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(rl[mb][s], qh[s], acc, 0, 0, 0); acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(rh[mb][s], qh[s], acc, 0, 0, 0); }
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const int j0 = 32 * mb + 8 * g + 4 * h;
                if (!live || j0 >= a.ws) continue;
                f32x4 v;
                for (int e = 0; e < 4; ++e) v[e] = j0 + e < a.ws ? acc[4 * g + e] * 0.03f : 0.f;
                const f16x4 hi = __builtin_convertvector(v, f16x4);
                _Float16* d = (V & 8) ? Rp + grow * JT + axis * a.ws + j0 : row + HD + axis * a.ws + j0;
                if ((V & 2) && !(never && v[0] == 12345.f)) continue;
                if (quad) *(f16x4*)d = hi;
                else for (int p2 = 0; p2 < 2; ++p2) if (j0 + 2 * p2 + 1 < a.ws) *(f16x2*)(d + 2 * p2) = f16x2{hi[2 * p2], hi[2 * p2 + 1]};
            }
        }
    }
}


// V2: coalesced q loads staged through a wave-private LDS tile, register prefetch of the next block, 16-byte stores
// after one lane swap (V2 & 1: f16x3-like double traffic is not modelled; PLAIN only)
typedef unsigned u32x4a __attribute__((ext_vector_type(4), aligned(4)));
typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
template <int HD, int MB, bool XR>
__global__ __launch_bounds__(256) void relpos2(const float* __restrict__ Rh, const float* __restrict__ Rw, _Float16* __restrict__ Qp,
                                               Geom a, int hpg, int n_tasks, unsigned ws_magic) {
    constexpr int KS = HD / 16, PR = HD / 8, ST = PR + 1, NT = PR / 2;   // pieces per row, LDS row stride (pieces), loads per lane
    __shared__ u32x4 stage_all[4][32 * ST];
    const int lane = threadIdx.x & 63, c = lane & 31, h = lane >> 5;
    u32x4* stage = stage_all[threadIdx.x >> 6];
    int bid = blockIdx.x;
    if (XR) { const int nb = gridDim.x, q8 = nb >> 3, r8 = nb & 7, x = bid & 7; bid = (x < r8 ? x * (q8 + 1) : r8 * (q8 + 1) + (x - r8) * q8) + (bid >> 3); }
    int task = bid * 4 + (threadIdx.x >> 6);
    if (task >= n_tasks) return;
    const int n_hg = a.heads / hpg;
    const int hg = task % n_hg; task /= n_hg;
    const int r = task % a.ws; task /= a.ws;
    const int axis = task & 1, wb = task >> 1;
    const int q_row = a.DQ;
    f16x8 rh[MB][KS], rl[MB][KS];
    const float* tab = (axis ? Rw : Rh) + (size_t)r * a.ws * HD + 8 * h;
    for (int mb = 0; mb < MB; ++mb) {
        const int j = 32 * mb + c;
#pragma unroll
        for (int s = 0; s < KS; ++s) {
            f32x4 v0 = {0.f, 0.f, 0.f, 0.f}, v1 = v0;
            if (j < a.ws) { v0 = *(const f32x4*)(tab + (size_t)j * HD + 16 * s); v1 = *(const f32x4*)(tab + (size_t)j * HD + 16 * s + 4); }
            f16x4 h0 = __builtin_convertvector(v0 * 256.f, f16x4), h1 = __builtin_convertvector(v1 * 256.f, f16x4);
            f16x4 l0 = __builtin_convertvector(v0 * 256.f - __builtin_convertvector(h0, f32x4), f16x4);
            f16x4 l1 = __builtin_convertvector(v1 * 256.f - __builtin_convertvector(h1, f32x4), f16x4);
            rh[mb][s] = cat(h0, h1); rl[mb][s] = cat(l0, l1);
        }
    }
    const int n_rho = a.ws * hpg;
    const size_t grp0 = (size_t)(wb * a.heads + hg * hpg) * a.Npad;
    auto row_of = [&](int rho) -> size_t {   // Q' row of the rho-th (head, position) of this line
        const int hl = __umulhi((unsigned)rho, ws_magic), i = rho - hl * a.ws;
        return grp0 + (size_t)hl * a.Npad + (axis ? i * a.ws + r : r * a.ws + i);
    };
    u32x4 pf[NT];
    auto fetch = [&](int rho0) {
#pragma unroll
        for (int t = 0; t < NT; ++t) {
            const int p = lane + 64 * t, cr = p / PR, pc = p - cr * PR;
            pf[t] = u32x4{0, 0, 0, 0};
            if (rho0 + cr < n_rho) pf[t] = *(const u32x4*)(Qp + row_of(rho0 + cr) * q_row + 8 * pc);
        }
    };
    fetch(0);
    for (int rho0 = 0; rho0 < n_rho; rho0 += 32) {
#pragma unroll
        for (int t = 0; t < NT; ++t) {
            const int p = lane + 64 * t, cr = p / PR, pc = p - cr * PR;
            stage[cr * ST + pc] = pf[t];
        }
        if (rho0 + 32 < n_rho) fetch(rho0 + 32);
        f16x8 qh[KS];
#pragma unroll
        for (int s = 0; s < KS; ++s) qh[s] = __builtin_bit_cast(f16x8, stage[c * ST + 2 * s + h]);
        const int rho = rho0 + c;
        const bool live = rho < n_rho;
        _Float16* row = Qp + row_of(rho) * q_row;
#pragma unroll
        for (int mb = 0; mb < MB; ++mb) {
            f32x16 acc;
            for (int e = 0; e < 16; ++e) acc[e] = 0.f;
#pragma unroll
            for (int s = 0; s < KS; ++s) { acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(rl[mb][s], qh[s], acc, 0, 0, 0); acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(rh[mb][s], qh[s], acc, 0, 0, 0); }
            u32x2 grp[4];
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                f32x4 v;
                for (int e = 0; e < 4; ++e) v[e] = 32 * mb + 8 * g + 4 * h + e < a.ws ? acc[4 * g + e] * 0.03f : 0.f;
                grp[g] = __builtin_bit_cast(u32x2, __builtin_convertvector(v, f16x4));
            }
#pragma unroll
            for (int p = 0; p < 2; ++p) {
                const u32x2 send = h ? grp[2 * p] : grp[2 * p + 1];
                const u32x2 recv = {(unsigned)__shfl_xor((int)send[0], 32), (unsigned)__shfl_xor((int)send[1], 32)};
                const u32x4 out = h ? u32x4{recv[0], recv[1], grp[2 * p + 1][0], grp[2 * p + 1][1]} : u32x4{grp[2 * p][0], grp[2 * p][1], recv[0], recv[1]};
                const int jb = 32 * mb + 16 * p + 8 * h, cnt = a.ws - jb;
                if (!live || cnt <= 0) continue;
                unsigned* d = (unsigned*)(row + HD + axis * a.ws + jb);
                if (cnt >= 8) *(u32x4a*)d = out;
                else for (int w2 = 0; w2 < 4; ++w2) if (2 * w2 + 1 < cnt) d[w2] = out[w2];
            }
        }
    }
}
template <int MB, bool XR>
float run2(const float* Rh, const float* Rw, _Float16* Qp, Geom a, int hpg) {
    const int n_tasks = a.B * a.nw * a.nw * 2 * a.ws * (a.heads / hpg);
    const unsigned magic = (unsigned)(((1ull << 32) + a.ws - 1) / a.ws);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    auto go = [&] { hipLaunchKernelGGL((relpos2<80, MB, XR>), dim3((n_tasks + 3) / 4), dim3(256), 0, 0, Rh, Rw, Qp, a, hpg, n_tasks, magic); };
    go(); hipDeviceSynchronize();
    hipEventRecord(e0); for (int i = 0; i < 10; ++i) go(); hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1); return ms / 10 * 1000;
}
__global__ void fill_rand(_Float16* p, size_t n, unsigned seed) {
    for (size_t i = blockIdx.x * 256ull + threadIdx.x; i < n; i += 256ull * gridDim.x) {
        unsigned x = (unsigned)i * 2654435761u + seed; x ^= x >> 15; x *= 2246822519u; x ^= x >> 13;
        p[i] = _Float16(float(int(x & 1023) - 512) * (1.f / 256.f));
    }
}
__global__ void fill_randf(float* p, size_t n, unsigned seed) {
    for (size_t i = blockIdx.x * 256ull + threadIdx.x; i < n; i += 256ull * gridDim.x) {
        unsigned x = (unsigned)i * 2654435761u + seed; x ^= x >> 15; x *= 2246822519u; x ^= x >> 13;
        p[i] = float(int(x & 65535) - 32768) * (1.f / 65536.f);
    }
}
__global__ void diff_count(const unsigned* a, const unsigned* b, size_t n, unsigned long long* out) {
    unsigned long long d = 0, nz = 0;
    for (size_t i = blockIdx.x * 256ull + threadIdx.x; i < n; i += 256ull * gridDim.x) { d += a[i] != b[i]; nz += a[i] != 0; }
    atomicAdd(out, d); atomicAdd(out + 1, nz);
}

template <int MB, int V>
float run(const float* Rh, const float* Rw, _Float16* Qp, _Float16* Rp, Geom a, int hpg) {
    const int n_tasks = a.B * a.nw * a.nw * ((V & 4) ? 1 : 2) * a.ws * (a.heads / hpg);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    auto go = [&] { hipLaunchKernelGGL((relpos<80, MB, V>), dim3((n_tasks + 3) / 4), dim3(256), 0, 0, Rh, Rw, Qp, Rp, a, hpg, n_tasks, 0); };
    go(); hipDeviceSynchronize();
    hipEventRecord(e0); for (int i = 0; i < 10; ++i) go(); hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1); return ms / 10 * 1000;
}
int main() {
    Geom w = {12, 14, 5, 16, 224, 112}, g = {12, 64, 1, 16, 4096, 208};
    size_t qbytes = (size_t)12 * 25 * 16 * 224 * 224 * 2 * 2;   // the product's allocation (2 DQ pitch)
    size_t qg = (size_t)12 * 16 * 4096 * 208 * 2 * 2;
    if (qg > qbytes) qbytes = qg;
    _Float16 *Qp, *Rp; float *Rh, *Rw;
    hipMalloc(&Qp, qbytes); hipMalloc(&Rp, qbytes); hipMalloc(&Rh, 64 * 64 * 80 * 4); hipMalloc(&Rw, 64 * 64 * 80 * 4);
    hipMemset(Qp, 0, qbytes); hipMemset(Rp, 0, qbytes); hipMemset(Rh, 0, 64 * 64 * 80 * 4); hipMemset(Rw, 0, 64 * 64 * 80 * 4);
    printf("window (12 images, 300 windows x 16 heads x 196 tokens), us per launch\n");
#define W(V) printf("  V=%2d %8.1f\n", V, run<1, V>(Rh, Rw, Qp, Rp, w, 16));
    W(0) W(1) W(2) W(3) W(4) W(8) W(9) W(10)
    printf("global (12 x 16 heads x 4096 tokens)\n");
#define G(V) printf("  V=%2d %8.1f\n", V, run<2, V>(Rh, Rw, Qp, Rp, g, 4));
    G(0) G(1) G(2) G(3) G(8)

    {   // V2 against V0 on random data: the same bits in every Q' word
        _Float16* Q2; hipMalloc(&Q2, qbytes);
        unsigned long long* cnt; hipMalloc(&cnt, 16);
        hipLaunchKernelGGL(fill_randf, dim3(1024), dim3(256), 0, 0, Rh, (size_t)64 * 64 * 80, 1u);
        hipLaunchKernelGGL(fill_randf, dim3(1024), dim3(256), 0, 0, Rw, (size_t)64 * 64 * 80, 2u);
        for (int which = 0; which < 2; ++which) {
            Geom a = which ? g : w;
            hipLaunchKernelGGL(fill_rand, dim3(4096), dim3(256), 0, 0, Qp, qbytes / 2, 7u);
            hipMemcpy(Q2, Qp, qbytes, hipMemcpyDeviceToDevice);
            if (which) { run<2, 0>(Rh, Rw, Qp, Rp, a, 4); printf("global V2 %8.1f us\n", run2<2, false>(Rh, Rw, Q2, a, 4)); }
            else { run<1, 0>(Rh, Rw, Qp, Rp, a, 16); printf("window V2 %8.1f us\n", run2<1, false>(Rh, Rw, Q2, a, 16)); }
            if (which) { printf("global V2 xr %8.1f  hpg2 %8.1f  hpg2 xr %8.1f  hpg8 xr %8.1f\n", run2<2, true>(Rh, Rw, Q2, a, 4), run2<2, false>(Rh, Rw, Q2, a, 2), run2<2, true>(Rh, Rw, Q2, a, 2), run2<2, true>(Rh, Rw, Q2, a, 8)); }
            else { printf("window V2 xr %8.1f  hpg8 %8.1f  hpg8 xr %8.1f  hpg4 xr %8.1f\n", run2<1, true>(Rh, Rw, Q2, a, 16), run2<1, false>(Rh, Rw, Q2, a, 8), run2<1, true>(Rh, Rw, Q2, a, 8), run2<1, true>(Rh, Rw, Q2, a, 4)); }
            hipMemset(cnt, 0, 16);
            hipLaunchKernelGGL(diff_count, dim3(4096), dim3(256), 0, 0, (const unsigned*)Qp, (const unsigned*)Q2, qbytes / 4, cnt);
            unsigned long long hc[2]; hipMemcpy(hc, cnt, 16, hipMemcpyDeviceToHost);
            printf("  differing words %llu of %llu non-zero\n", hc[0], hc[1]);
        }
    }
    return 0;
}
