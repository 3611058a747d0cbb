/* chain_probe2.hip -- what the feedback recurrence's step (wosc.h:273-310, lanes = chains) is made of on gfx950 (round 4):
 *   hipcc --offload-arch=gfx950 -O3 -ffp-contract=off -I saugns_amd/csrc tools/chain_probe2.hip -o /tmp/chain_probe2
 * The bare loop of tools/chain_probe.hip with (ACT) 64, 32 or 16 active lanes per wave (the rest masked off), with (MODE)
 * 0 the table entry as it is staged today (f64 [c3, c2] + f32 [c1, c0]), 1 wide entries ([c1, c0] as f64, one address),
 * 2 no table read at all (one fixed entry: the arithmetic chain alone), 3 the entry of the *previous* step's index (the
 * read off the dependent chain: what a perfect index prediction would give). ns per sample step each. */
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <vector>
#include "sau_dev_math.h"
using namespace saudev;
typedef double __attribute__((ext_vector_type(2))) f64x2;
typedef float __attribute__((ext_vector_type(2))) f32x2;
typedef const f64x2 __attribute__((address_space(3))) *lds_f64x2;
typedef const f32x2 __attribute__((address_space(3))) *lds_f32x2;

template <int ACT, int MODE>
__global__ void __launch_bounds__(64) probe(const HerpC23 *g23, const HerpC01 *g01, const uint32_t *base,
		const float *pma, float *out, uint32_t stride, uint32_t n, float dscale, float doff) {
	extern __shared__ __align__(16) unsigned char lds[];
	{
		f64x2 *t23 = (f64x2 *)lds;
		for (uint32_t i = threadIdx.x; i < WAVE_LEN; i += 64) { f64x2 v; v.x = g23[i].c3; v.y = g23[i].c2; t23[i] = v; }
		if (MODE == 1) {
			f64x2 *t01 = (f64x2 *)(lds + 32768);
			for (uint32_t i = threadIdx.x; i < WAVE_LEN; i += 64) { f64x2 v; v.x = g01[i].c1; v.y = g01[i].c0; t01[i] = v; }
		} else {
			f32x2 *t01 = (f32x2 *)(lds + 32768);
			for (uint32_t i = threadIdx.x; i < WAVE_LEN; i += 64) { f32x2 v; v.x = g01[i].c1; v.y = g01[i].c0; t01[i] = v; }
		}
	}
	__syncthreads();
	if ((int)threadIdx.x >= ACT) return;
	const uint32_t chain = blockIdx.x * ACT + threadIdx.x;
	const uint4 *bp = (const uint4 *)(base + (size_t)chain * stride);
	const float4 *pp = (const float4 *)(pma + (size_t)chain * stride);
	float4 *op = (float4 *)(out + (size_t)chain * stride);
	const uint32_t tab = (uint32_t)(uintptr_t)lds;
	uint32_t prev_phase = 0, prev_ind = 0; double prev_Is = 0; float prev_s = 0, fb_s = 0;
	constexpr int U = 8; /* batches of 4 steps in flight: the rows' HBM round trip stays off the measured chain */
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
			uint32_t ind = phase >> SLEN_BITS;
			if (MODE == 2) ind = 7;
			if (MODE == 3) { const uint32_t x = prev_ind; prev_ind = ind; ind = x; }
			double c3, c2, c1, c0;
			if (MODE == 1) {
				const uint32_t a = tab + (ind << 4);
				const f64x2 hi = *(lds_f64x2)(uintptr_t)a; const f64x2 lo = *(lds_f64x2)(uintptr_t)(a + 32768);
				c3 = hi.x; c2 = hi.y; c1 = lo.x; c0 = lo.y;
			} else {
				const f64x2 hi = *(lds_f64x2)(uintptr_t)(tab + (ind << 4)); const f32x2 lo = *(lds_f32x2)(uintptr_t)(tab + 32768 + (ind << 3));
				c3 = hi.x; c2 = hi.y; c1 = (double)lo.x; c0 = (double)lo.y;
			}
			const double x = (double)(phase & (SLEN - 1));
			const double Isv = ((c3 * x + c2) * x + c1) * x + c0;
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

template <int ACT, int MODE>
static void run(const char *what, uint32_t chains, const HerpC23 *d23, const HerpC01 *d01, const uint32_t *base, const float *pma, float *out,
		uint32_t stride, uint32_t n) {
	const size_t lds = 65536;
	hipFuncSetAttribute((const void *)probe<ACT, MODE>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
	hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
	float best = 1e9f;
	for (int rep = 0; rep < 3; ++rep) {
		hipEventRecord(a);
		hipLaunchKernelGGL((probe<ACT, MODE>), dim3(chains / ACT), dim3(64), lds, 0, d23, d01, base, pma, out, stride, n, 6.8e8f, 0.f);
		hipEventRecord(b); hipEventSynchronize(b);
		float ms; hipEventElapsedTime(&ms, a, b);
		if (rep && ms < best) best = ms;
	}
	printf("%-52s %2d lanes/wave, %4u waves: %6.1f ns per step\n", what, ACT, chains / ACT, best * 1e6 / n);
}

int main(int argc, char **argv) {
	const uint32_t n = 65536, chains = argc > 1 ? atoi(argv[1]) : 4096, stride = n;
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
	run<64, 0>("entry as staged today (f64 pair + f32 pair)", chains, d23, d01, base, pma, out, stride, n);
	run<32, 0>("entry as staged today (f64 pair + f32 pair)", chains, d23, d01, base, pma, out, stride, n);
	run<16, 0>("entry as staged today (f64 pair + f32 pair)", chains, d23, d01, base, pma, out, stride, n);
	run<64, 1>("wide entry (two f64 pairs, one address)", chains, d23, d01, base, pma, out, stride, n);
	run<32, 1>("wide entry (two f64 pairs, one address)", chains, d23, d01, base, pma, out, stride, n);
	run<16, 1>("wide entry (two f64 pairs, one address)", chains, d23, d01, base, pma, out, stride, n);
	run<64, 2>("no table read (arithmetic chain alone)", chains, d23, d01, base, pma, out, stride, n);
	run<16, 2>("no table read (arithmetic chain alone)", chains, d23, d01, base, pma, out, stride, n);
	run<64, 3>("entry read off the chain (previous step's index)", chains, d23, d01, base, pma, out, stride, n);
	run<32, 3>("entry read off the chain (previous step's index)", chains, d23, d01, base, pma, out, stride, n);
	std::vector<float> ho(8); hipMemcpy(ho.data(), out + 1000, 32, hipMemcpyDeviceToHost);
	printf("out[1000..]: %g %g %g %g\n", ho[0], ho[1], ho[2], ho[3]);
	return 0;
}
