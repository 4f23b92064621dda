// Dev harness: f16x3 GEMM vs f32-MFMA GEMM — speed and accuracy against an fp64 host reference.
#include "../pope_amd/csrc/kernels.h"
#include <cmath>
#include <cstdio>
#include <vector>

int pope_lab_gemm_f16x3(const GemmParams& g, int lab, hipStream_t stream);
static int g_lab = 0;
static int lab_fn(const GemmParams& g, hipStream_t s) { return pope_lab_gemm_f16x3(g, g_lab, s); }

static float timeit(int (*fn)(const GemmParams&, hipStream_t), const GemmParams& g) {
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    float best = 1e9;
    for (int r = 0; r < 4; ++r) {
        hipEventRecord(a); fn(g, 0); hipEventRecord(b); hipEventSynchronize(b);
        float ms; hipEventElapsedTime(&ms, a, b); if (ms < best) best = ms;
    }
    return best;
}

int main() {
    const int M = 64 * 1531;
    size_t na = size_t(M) * 1536, nw = size_t(1536) * 1536, nc = size_t(M) * 1536;
    std::vector<float> ha(na), hw(nw), hb(1536);
    unsigned s = 1;
    auto rnd = [&]() { s = s * 1664525u + 1013904223u; return (s >> 8) * (1.0f / 8388608.0f) - 1.0f; };
    for (auto& v : ha) v = 2.5f * rnd() * rnd() * 2.f;   // activations: heavy-ish tails, |x| < 5
    for (auto& v : hw) v = 0.08f * rnd();
    for (auto& v : hb) v = 0.1f * rnd();
    float *A, *W, *C, *Bv; hipMalloc(&A, na * 4); hipMalloc(&W, nw * 4); hipMalloc(&C, nc * 4); hipMalloc(&Bv, 1536 * 4);
    hipMemcpy(A, ha.data(), na * 4, hipMemcpyHostToDevice); hipMemcpy(W, hw.data(), nw * 4, hipMemcpyHostToDevice);
    hipMemcpy(Bv, hb.data(), 1536 * 4, hipMemcpyHostToDevice);
    struct { const char* name; int N, K; } shapes[] = {{"qkv  N1152 K384 ", 1152, 384}, {"proj N384  K384 ", 384, 384},
                                                       {"fc1  N1536 K384 ", 1536, 384}, {"fc2  N384  K1536", 384, 1536}};
    void *Ah, *Al, *Wh, *Wl;
    hipMalloc(&Ah, na * 4); hipMalloc(&Wh, nw * 4); Al = Wl = nullptr; (void)Al; (void)Wl;
    std::vector<float> hc(size_t(256) * 1536);
    for (auto& sh : shapes) {
        GemmParams g = {};
        g.A = A; g.W = W; g.bias = Bv; g.C = C; g.M = M; g.N = sh.N; g.K = sh.K; g.lda = sh.K; g.ldw = sh.K; g.ldc = sh.N;
        g.epilogue = EPI_BIAS;
        const double gf = 2.0 * M * sh.N * sh.K / 1e9;
        double err[2] = {0, 0}, ref_rms = 0;
        float t[2];
        for (int v = 0; v < 2; ++v) {
            auto fn = v ? pope_launch_gemm_nt_f16x3 : pope_launch_gemm_nt_f32;
            t[v] = timeit(fn, g);
            hipMemcpy(hc.data(), C, size_t(256) * sh.N * 4, hipMemcpyDeviceToHost);   // first 256 rows
            double e = 0, rr = 0;
            for (int m = 0; m < 256; m += 5)
                for (int n = 0; n < sh.N; n += 7) {
                    double acc = hb[n];
                    for (int k = 0; k < sh.K; ++k) acc += double(ha[size_t(m) * sh.K + k]) * double(hw[size_t(n) * sh.K + k]);
                    e = fmax(e, fabs(acc - hc[size_t(m) * sh.N + n])); rr += acc * acc;
                }
            err[v] = e; ref_rms = sqrt(rr / ((256 / 5 + 1) * (sh.N / 7 + 1)));
        }
        float tl[7] = {0};
        for (int l = 1; l <= 6; ++l) { g_lab = l; tl[l] = timeit(lab_fn, g); }
        printf("   ablations (ms): noload %.3f | nosplit/ldswrite %.3f | mfma+ldsread only %.3f | nomfma %.3f | nomfma+noload %.3f | nomfma+nosplit %.3f\n",
               tl[1], tl[2], tl[3], tl[4], tl[5], tl[6]);
        {   // planes variant: operands pre-split once (as LayerNorm / the weight loader would)
            pope_launch_split_planes(A, Ah, M, sh.K, K_PLANES_ACT_SCALE, 0);
            pope_launch_split_planes(W, Wh, sh.N, sh.K, K_PLANES_W_SCALE, 0);
            GemmParams gp = g; gp.a_pl = Ah; gp.w_pl = Wh;
            hipMemset(C, 0, size_t(256) * sh.N * 4);
            float tp = timeit(pope_launch_gemm_nt_f16x3_planes, gp);
            hipMemcpy(hc.data(), C, size_t(256) * sh.N * 4, hipMemcpyDeviceToHost);
            double e = 0;
            for (int m = 0; m < 256; m += 5)
                for (int n = 0; n < sh.N; n += 7) {
                    double acc = hb[n];
                    for (int k = 0; k < sh.K; ++k) acc += double(ha[size_t(m) * sh.K + k]) * double(hw[size_t(n) * sh.K + k]);
                    e = fmax(e, fabs(acc - hc[size_t(m) * sh.N + n]));
                }
            printf("   PLANES kernel: %.3f ms (%.1f TF-eq) maxerr %.2e\n", tp, gf / tp, e);
#ifdef X3_STAMPS
            if (sh.K == 384) {
                const size_t nd = size_t(512) * 16 * 16;
                unsigned long long* dbg; hipMalloc(&dbg, nd * 8); hipMemset(dbg, 0, nd * 8);
                gp.posb = reinterpret_cast<const float*>(dbg);
                hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
                hipEventRecord(e0); pope_launch_gemm_nt_f16x3_planes(gp, 0); hipEventRecord(e1); hipDeviceSynchronize();
                float ems; hipEventElapsedTime(&ems, e0, e1);
                std::vector<unsigned long long> hd(nd);
                hipMemcpy(hd.data(), dbg, nd * 8, hipMemcpyDeviceToHost);
                {
                    unsigned long long t0 = ~0ull;
                    for (int b = 0; b < 512; ++b) if (hd[(b * 16) * 16 + 14] && hd[(b * 16) * 16 + 14] < t0) t0 = hd[(b * 16) * 16 + 14];
                    printf("   stamped launch: %.1f us by events; blocks [first K-step, last epilogue end] us after the earliest stamp:", ems * 1e3);
                    for (int b = 0; b < 512; b += 37) {
                        unsigned long long last_rt = 0; int nt = 0;
                        for (int t = 0; t < 16; ++t) if (hd[(b * 16 + t) * 16 + 15]) { last_rt = hd[(b * 16 + t) * 16 + 15]; ++nt; }
                        printf(" b%d[%.1f,%.1f;%dt]", b, (hd[(b * 16) * 16 + 14] - t0) / 100.0, (last_rt - t0) / 100.0, nt);
                    }
                    printf("\n");
                }
                for (int b : {0, 40}) {
                    printf("   block %2d: per tile [K-loop | epilogue | gap to next tile]:", b);
                    for (int t = 0; t < 15; ++t) {
                        auto* q = &hd[(b * 16 + t) * 16];
                        if (!q[13]) break;
                        auto* qn = &hd[(b * 16 + t + 1) * 16];
                        printf(" [%llu|%llu|%lld]", q[12] - q[0], q[13] - q[12], qn[0] ? (long long)(qn[0] - q[13]) : -1LL);
                    }
                    auto* q0 = &hd[(b * 16) * 16];
                    unsigned long long last = 0, last_rt = 0;
                    for (int t = 0; t < 16; ++t) if (hd[(b * 16 + t) * 16 + 13]) { last = hd[(b * 16 + t) * 16 + 13]; last_rt = hd[(b * 16 + t) * 16 + 15]; }
                    const double us = (last_rt - q0[14]) / 100.0;   // s_memrealtime ticks at 100 MHz
                    printf("  total first K-step..last epilogue %llu cycles in %.1f us -> shader clock %.3f GHz\n", last - q0[0], us,
                           (last - q0[0]) / us / 1e3);
                }
            }
#endif

        }
        printf("%s f32-mfma %.3f ms (%.1f TF) maxerr %.2e | f16x3 %.3f ms (%.1f TF-eq) maxerr %.2e | out rms %.2f\n", sh.name, t[0],
               gf / t[0], err[0], t[1], gf / t[1], err[1], ref_rms);
    }
    return 0;
}
