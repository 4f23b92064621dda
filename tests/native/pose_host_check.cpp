// TEST INFRASTRUCTURE: host build of pope_amd/csrc/pose_math.h (the arithmetic pose.hip runs one hypothesis per thread)
// behind a C interface for ctypes, so that tests/test_pose_cpu.py can hold it against oracle/pose_ref.py without a GPU.
// Built by the test into tests/native/_build/ (git-ignored); never part of the product.
#include "../../pope_amd/csrc/pose_math.h"

extern "C" {
int host_five_point(const double* x0, const double* x1, double* E_out /* [10][9] */) {
    return pose::five_point(x0, x1, reinterpret_cast<double (*)[9]>(E_out));
}
void host_sample_indices(unsigned long long seed, unsigned h, unsigned n, int* picks) { pose::sample_indices(seed, h, n, picks); }
int host_sturm_roots(const double* c, int d, double lo, double hi, double* out) { return pose::sturm_roots(c, d, lo, hi, out); }
void host_decompose(const double* E, double* R1, double* R2, double* t) { pose::decompose_essential(E, R1, R2, t); }
int host_cheirality(const double* R, const double* t, double ax, double ay, double bx, double by, double dist) {
    return pose::cheirality(R, t, ax, ay, bx, by, dist) ? 1 : 0;
}
double host_sampson(const double* E, double ax, double ay, double bx, double by) { return pose::sampson(E, ax, ay, bx, by); }
int host_update_num_iters(double conf, double ratio, int max_iters) { return pose::update_num_iters(conf, ratio, max_iters); }
}
