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
                unsigned long long* dbg; hipMalloc(&dbg, 64 * 4 * 16 * 8); hipMemset(dbg, 0, 64 * 4 * 16 * 8);
                gp.posb = reinterpret_cast<const float*>(dbg);
                pope_launch_gemm_nt_f16x3_planes(gp, 0); hipDeviceSynchronize();
                std::vector<unsigned long long> hd(64 * 4 * 16);
                hipMemcpy(hd.data(), dbg, hd.size() * 8, hipMemcpyDeviceToHost);
                for (int b : {0, 9, 40}) for (int t = 1; t < 3; ++t) {
                    auto* q = &hd[(b * 4 + t) * 16];
                    printf("   block %2d tile %d: K-steps(0..9):", b, t);
                    unsigned long long prev = q[0];
                    for (int k = 1; k <= 10; ++k) { printf(" %llu", q[k] - prev); prev = q[k]; }
                    printf(" | last two K-steps %llu | epilogue %llu | whole tile %llu\n", q[12] - q[10], q[13] - q[12], q[13] - q[0]);
                }
            }
#endif

        }
        printf("%s f32-mfma %.3f ms (%.1f TF) maxerr %.2e | f16x3 %.3f ms (%.1f TF-eq) maxerr %.2e | out rms %.2f\n", sh.name, t[0],
               gf / t[0], err[0], t[1], gf / t[1], err[1], ref_rms);
    }
    return 0;
}
