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

/* rint64() (sau_dev_math.h: llrintf with the host's out-of-range answer) for EVERY f32 bit pattern against an independent
 * formulation -- the value rounded in f64 and taken apart by hand -- in two lane orders: consecutive patterns (a wave's lanes
 * alike in magnitude: the wave-uniform short form below 2^31, or the long form) and scattered ones (both forms' lanes in one
 * wave). franssgauss32()'s scaled conversions against the reference's double form, for every 32-bit integer, ride along. */
__device__ __forceinline__ long long kat_rint64_ref(float x) {
	if (!(fabsf(x) < 0x1p63f)) return (long long)0x8000000000000000ull; /* (NaN too) */
	const double d = rint((double)x); /* round-half-even, exact: an f32 has 24 significant bits */
	const double a = fabs(d);
	const double hi = floor(a * 0x1p-32);
	const unsigned long long mag = ((unsigned long long)(uint32_t)hi << 32) | (unsigned long long)(uint32_t)(a - hi * 0x1p32);
	return d < 0 ? (long long)(0ull - mag) : (long long)mag;
}
__global__ void kat_rint64_kernel(int scattered, unsigned long long *mismatches, uint32_t *first_bad) {
	unsigned long long bad = 0;
	const uint32_t tid = blockIdx.x * blockDim.x + threadIdx.x, n = gridDim.x * blockDim.x; /* n divides 2^32 */
	for (uint32_t k = 0; k < (uint32_t)(0x100000000ull / n); ++k) {
		uint32_t bits = k * n + tid;
		if (scattered) bits *= 0x9e3779b1u; /* (odd: a permutation of the patterns; neighbouring lanes far apart) */
		const float x = bits_f(bits);
		if (rint64(x) != kat_rint64_ref(x)) { ++bad; atomicMin(first_bad, bits); }
		const int32_t s0 = (int32_t)bits;
		if (f_bits((float)s0 * 0x1p-32f) != f_bits((float)((double)s0 * 0x1p-32))) { ++bad; atomicMin(first_bad, bits); }
	}
	if (bad) atomicAdd(mismatches, bad);
}
bool kat_rint64(int scattered, unsigned long long *mismatches, uint32_t *first_bad) {
	unsigned long long *d_m = nullptr;
	uint32_t *d_f = nullptr;
	bool ok = hipMalloc((void **)&d_m, sizeof *d_m) == hipSuccess && hipMalloc((void **)&d_f, sizeof *d_f) == hipSuccess;
	if (ok) {
		const uint32_t none = 0xffffffffu;
		ok = hipMemset(d_m, 0, sizeof *d_m) == hipSuccess && hipMemcpy(d_f, &none, sizeof none, hipMemcpyHostToDevice) == hipSuccess;
	}
	if (ok) {
		hipLaunchKernelGGL(kat_rint64_kernel, dim3(4096), dim3(256), 0, 0, scattered, d_m, d_f);
		ok = hipDeviceSynchronize() == hipSuccess &&
			hipMemcpy(mismatches, d_m, sizeof *d_m, hipMemcpyDeviceToHost) == hipSuccess &&
			hipMemcpy(first_bad, d_f, sizeof *d_f, hipMemcpyDeviceToHost) == hipSuccess;
	}
	if (d_m) (void)hipFree(d_m);
	if (d_f) (void)hipFree(d_f);
	return ok;
}
