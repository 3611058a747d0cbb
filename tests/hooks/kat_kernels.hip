/* kat_kernels.hip -- TEST INFRASTRUCTURE (tests/hooks/libsaugns_amd_hooks.so only): known-answer probes of the shared
 * arithmetic (saugns_amd/csrc/sau_dev_math.h) as hipcc compiles it for gfx950, with the product's flags. Until round 4
 * these kernels sat in the product library; VERDICT r04 item 9 moved them here. */
#include <hip/hip_runtime.h>
#include "../../saugns_amd/csrc/sau_dev_ops.h"
#include <stdint.h>
#include "../../saugns_amd/csrc/k_wave_scan.h"

using namespace saudev;

/* div_diff_scale(a, b) against IEEE a / b for every f32 b with 1 <= |b| <= 2^31 (a superset of the rounded integers the
 * differentiator divides by). Thread t takes the bit patterns t, t + stride, ... */
__global__ void kat_div_kernel(float a, int variant, unsigned long long *mismatches, uint32_t *first_bad) {
	const uint32_t lo = 0x3f800000u, hi = 0x4f000000u; /* 1.0f .. 2^31 */
	unsigned long long bad = 0;
	for (uint32_t bits = lo + blockIdx.x * blockDim.x + threadIdx.x; bits <= hi; bits += gridDim.x * blockDim.x) {
		for (int sgn = 0; sgn < 2; ++sgn) {
			const float b = bits_f(bits | (sgn ? 0x80000000u : 0u));
			const float want = __fdiv_rn(a, b);
			float got;
			if (variant == 0) {
				got = div_diff_scale(a, b);
			} else { /* no correction: the probe must be able to see this fail */
				got = a * __builtin_amdgcn_rcpf(b);
			}
			if (f_bits(want) != f_bits(got)) { ++bad; atomicMin(first_bad, bits); }
		}
		if (bits > hi - gridDim.x * blockDim.x) break; /* no wrap past the last pattern */
	}
	if (bad) atomicAdd(mismatches, bad);
}

/* one block evaluates a line for `len` samples exactly as the kernels' ST_LINE step does */
__global__ void kat_line_kernel(LineState st, uint32_t len, const float *mul, float *out, LineState *st_out) {
	LineState ls = st;
	LineBlock lb = line_begin(ls, len, mul != nullptr, mul ? mul[0] : 0.f, lattice_none(), 0);
	for (uint32_t j = threadIdx.x; j < len; j += blockDim.x)
		out[j] = line_value(lb, j, mul ? mul[j] : 1.f);
	if (threadIdx.x == 0) *st_out = ls;
}

bool kat_div(float a, int variant, unsigned long long *mismatches, uint32_t *first_bad) {
	unsigned long long *d_m = nullptr;
	uint32_t *d_f = nullptr;
	bool ok = hipMalloc((void **)&d_m, sizeof *d_m) == hipSuccess && hipMalloc((void **)&d_f, sizeof *d_f) == hipSuccess;
	if (ok) {
		const uint32_t none = 0xffffffffu;
		ok = hipMemset(d_m, 0, sizeof *d_m) == hipSuccess &&
			hipMemcpy(d_f, &none, sizeof none, hipMemcpyHostToDevice) == hipSuccess;
	}
	if (ok) {
		hipLaunchKernelGGL(kat_div_kernel, dim3(4096), dim3(256), 0, 0, a, variant, d_m, d_f);
		ok = hipDeviceSynchronize() == hipSuccess &&
			hipMemcpy(mismatches, d_m, sizeof *d_m, hipMemcpyDeviceToHost) == hipSuccess &&
			hipMemcpy(first_bad, d_f, sizeof *d_f, hipMemcpyDeviceToHost) == hipSuccess;
	}
	if (d_m) (void)hipFree(d_m);
	if (d_f) (void)hipFree(d_f);
	return ok;
}

bool kat_line(const LineState &st, uint32_t len, const float *mul, float *out, LineState *st_out) {
	float *d_mul = nullptr, *d_out = nullptr;
	LineState *d_st = nullptr;
	bool ok = hipMalloc((void **)&d_out, (len + 1) * sizeof(float)) == hipSuccess &&
		hipMalloc((void **)&d_st, sizeof(LineState)) == hipSuccess;
	if (ok && mul) {
		ok = hipMalloc((void **)&d_mul, (len + 1) * sizeof(float)) == hipSuccess &&
			hipMemcpy(d_mul, mul, len * sizeof(float), hipMemcpyHostToDevice) == hipSuccess;
	}
	if (ok) {
		hipLaunchKernelGGL(kat_line_kernel, dim3(1), dim3(256), 0, 0, st, len, d_mul, d_out, d_st);
		ok = hipDeviceSynchronize() == hipSuccess &&
			hipMemcpy(out, d_out, len * sizeof(float), hipMemcpyDeviceToHost) == hipSuccess &&
			hipMemcpy(st_out, d_st, sizeof(LineState), hipMemcpyDeviceToHost) == hipSuccess;
	}
	if (d_mul) (void)hipFree(d_mul);
	if (d_out) (void)hipFree(d_out);
	if (d_st) (void)hipFree(d_st);
	return ok;
}

/* the 64-bit wave scan and sum (k_wave_scan.h): one wave per 64 values */
__global__ void kat_scan64_kernel(const unsigned long long *in, unsigned long long *scan, unsigned long long *sum) {
	const uint32_t i = blockIdx.x * 64 + threadIdx.x;
	scan[i] = wave_incl_scan64_dpp(in[i]);
	const unsigned long long t = wave_sum64_dpp(in[i]);
	if (threadIdx.x == 0) sum[blockIdx.x] = t;
}
bool kat_scan64(const unsigned long long *in, unsigned long long *scan, unsigned long long *sum, uint32_t n_waves) {
	unsigned long long *d_in = nullptr, *d_scan = nullptr, *d_sum = nullptr;
	const size_t nb = (size_t)n_waves * 64 * sizeof(unsigned long long);
	bool ok = hipMalloc((void **)&d_in, nb) == hipSuccess && hipMalloc((void **)&d_scan, nb) == hipSuccess &&
		hipMalloc((void **)&d_sum, n_waves * sizeof(unsigned long long)) == hipSuccess &&
		hipMemcpy(d_in, in, nb, hipMemcpyHostToDevice) == hipSuccess;
	if (ok) {
		hipLaunchKernelGGL(kat_scan64_kernel, dim3(n_waves), dim3(64), 0, 0, d_in, d_scan, d_sum);
		ok = hipDeviceSynchronize() == hipSuccess && hipMemcpy(scan, d_scan, nb, hipMemcpyDeviceToHost) == hipSuccess &&
			hipMemcpy(sum, d_sum, n_waves * sizeof(unsigned long long), hipMemcpyDeviceToHost) == hipSuccess;
	}
	if (d_in) (void)hipFree(d_in);
	if (d_scan) (void)hipFree(d_scan);
	if (d_sum) (void)hipFree(d_sum);
	return ok;
}
