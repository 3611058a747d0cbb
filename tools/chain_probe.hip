/* chain_probe.hip -- latency probe for the feedback recurrence (wosc.h:273-310) with lanes = voices:
 * one wave runs 64 independent chains, base phases / amounts streamed from per-chain rows in HBM.
 *   hipcc --offload-arch=gfx950 -O3 -ffp-contract=off -I saugns_amd/csrc tools/chain_probe.hip -o /tmp/chain_probe
 * prints ns per sample step (all 64 chains advance one sample per step). */
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <vector>
#include "sau_dev_math.h"
using namespace saudev;

typedef const HerpC23 __attribute__((address_space(3))) *lds_c23_ptr;
typedef const HerpC01 __attribute__((address_space(3))) *lds_c01_ptr;

template <int U>
__global__ void __launch_bounds__(64) chain_kernel(const HerpC23 *g23, const HerpC01 *g01, const uint32_t *base,
		const float *pma, float *out, uint32_t stride, uint32_t n, float dscale, float doff) {
	extern __shared__ __align__(16) unsigned char lds[];
	HerpC23 *t23 = (HerpC23 *)lds;
	HerpC01 *t01 = (HerpC01 *)(lds + WAVE_LEN * sizeof(HerpC23));
	for (uint32_t i = threadIdx.x; i < WAVE_LEN; i += 64) { t23[i] = g23[i]; t01[i] = g01[i]; }
	__syncthreads();
	const uint32_t chain = blockIdx.x * 64 + threadIdx.x;
	const uint4 *bp = (const uint4 *)(base + (size_t)chain * stride);
	const float4 *pp = (const float4 *)(pma + (size_t)chain * stride);
	float4 *op = (float4 *)(out + (size_t)chain * stride);
	lds_c23_ptr l23 = (lds_c23_ptr)t23;
	lds_c01_ptr l01 = (lds_c01_ptr)t01;
	uint32_t prev_phase = 0; double prev_Is = 0; float prev_s = 0, fb_s = 0;
	uint4 bq[U]; float4 pq[U];
#pragma unroll
	for (int u = 0; u < U; ++u) { bq[u] = bp[u]; pq[u] = pp[u]; }
	for (uint32_t t = 0; t < n; t += 4 * U) {
		uint4 bn[U]; float4 pn[U];
		const uint32_t nx = (t + 4 * U < n) ? (t + 4 * U) / 4 : 0;
#pragma unroll
		for (int u = 0; u < U; ++u) { bn[u] = bp[nx + u]; pn[u] = pp[nx + u]; }
#pragma unroll
		for (int u = 0; u < U; ++u) {
			const uint32_t b4[4] = {bq[u].x, bq[u].y, bq[u].z, bq[u].w};
			const float p4[4] = {pq[u].x, pq[u].y, pq[u].z, pq[u].w};
			float s4[4];
#pragma unroll
			for (int j = 0; j < 4; ++j) {
				const float p = fb_s * p4[j];
				const uint32_t ofs = (uint32_t)__double2loint((double)p + 0x1.8p21);
				const uint32_t phase = b4[j] + ofs;
				const int32_t d = (int32_t)(phase - prev_phase);
				const uint32_t ind = phase >> SLEN_BITS;
				HerpC23 hi; HerpC01 lo;
				hi.c3 = l23[ind].c3; hi.c2 = l23[ind].c2; lo.c1 = l01[ind].c1; lo.c0 = l01[ind].c0;
				const double Isv = herp_poly(hi, lo, phase);
				const float sv_new = wosc_diff(Isv, prev_Is, d, dscale, doff);
				const bool hold = d == 0;
				const float sv = hold ? prev_s : sv_new;
				prev_Is = hold ? prev_Is : Isv;
				prev_phase = phase;
				prev_s = sv;
				s4[j] = sv;
				fb_s = (fb_s + sv) * 0.5f;
			}
			op[t / 4 + u] = make_float4(s4[0], s4[1], s4[2], s4[3]);
		}
#pragma unroll
		for (int u = 0; u < U; ++u) { bq[u] = bn[u]; pq[u] = pn[u]; }
	}
}

int main(int argc, char **argv) {
	const uint32_t n = 65536, waves = argc > 1 ? atoi(argv[1]) : 64, chains = waves * 64, stride = n;
	std::vector<float> lut(WAVE_LEN);
	for (uint32_t i = 0; i < WAVE_LEN; ++i) lut[i] = (float)(-cos(2 * M_PI * i / WAVE_LEN) * 0.159);
	std::vector<HerpC23> h23(WAVE_LEN); std::vector<HerpC01> h01(WAVE_LEN);
	for (uint32_t i = 0; i < WAVE_LEN; ++i) herp_coeffs(lut.data(), i, h23[i], h01[i]);
	HerpC23 *d23; HerpC01 *d01; uint32_t *base; float *pma, *out;
	hipMalloc(&d23, sizeof(HerpC23) * WAVE_LEN); hipMalloc(&d01, sizeof(HerpC01) * WAVE_LEN);
	hipMemcpy(d23, h23.data(), sizeof(HerpC23) * WAVE_LEN, hipMemcpyHostToDevice);
	hipMemcpy(d01, h01.data(), sizeof(HerpC01) * WAVE_LEN, hipMemcpyHostToDevice);
	hipMalloc(&base, (size_t)chains * stride * 4); hipMalloc(&pma, (size_t)chains * stride * 4); hipMalloc(&out, (size_t)chains * stride * 4);
	std::vector<uint32_t> hb((size_t)chains * stride); std::vector<float> hp((size_t)chains * stride);
	for (uint32_t c = 0; c < chains; ++c) for (uint32_t t = 0; t < n; ++t) {
		hb[(size_t)c * stride + t] = (uint32_t)((uint64_t)(t + 1) * (7791327u + c * 20551u));
		hp[(size_t)c * stride + t] = 0.3f + 0.1f * (c % 8);
	}
	hipMemcpy(base, hb.data(), hb.size() * 4, hipMemcpyHostToDevice);
	hipMemcpy(pma, hp.data(), hp.size() * 4, hipMemcpyHostToDevice);
	const size_t lds = WAVE_LEN * (sizeof(HerpC23) + sizeof(HerpC01));
	hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
	for (int U : {1, 2, 4}) {
		for (int rep = 0; rep < 2; ++rep) {
			hipEventRecord(a);
			if (U == 1) hipLaunchKernelGGL(chain_kernel<1>, dim3(waves), dim3(64), lds, 0, d23, d01, base, pma, out, stride, n, 6.8e8f, 0.f);
			if (U == 2) hipLaunchKernelGGL(chain_kernel<2>, dim3(waves), dim3(64), lds, 0, d23, d01, base, pma, out, stride, n, 6.8e8f, 0.f);
			if (U == 4) hipLaunchKernelGGL(chain_kernel<4>, dim3(waves), dim3(64), lds, 0, d23, d01, base, pma, out, stride, n, 6.8e8f, 0.f);
			hipEventRecord(b); hipEventSynchronize(b);
			float ms; hipEventElapsedTime(&ms, a, b);
			if (rep) printf("waves %u U %d: %.3f ms for %u samples -> %.1f ns per step; %.3e chain-samples/s\n", waves, U, ms, n, ms * 1e6 / n, (double)chains * n / (ms * 1e-3));
		}
	}
	std::vector<float> ho(8); hipMemcpy(ho.data(), out + 1000, 32, hipMemcpyDeviceToHost);
	printf("out[1000..]: %g %g %g %g\n", ho[0], ho[1], ho[2], ho[3]);
	return 0;
}
