/* k_wave_scan.h -- wave-wide sums by DPP (no LDS): part of hip_backend.hip through k_common.h (inside namespace sauhip), and,
 * for the known-answer probes, of tests/hooks/kat_kernels.hip. Device code only. */
#ifndef SAU_K_WAVE_SCAN_H
#define SAU_K_WAVE_SCAN_H
#include <stdint.h>

/* inclusive sum over the 64 lanes with DPP moves (no LDS): four shifts inside
 * each row of 16, then the rows' totals passed on with row_bcast 15 and 31 */
__device__ __forceinline__ uint32_t wave_incl_scan_dpp(uint32_t v) {
#define SAU_DPP_ADD(ctrl, rmask) v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, ctrl, rmask, 0xf, false)
	v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x111 /* row_shr:1 */, 0xf, 0xf, true);
	v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x112 /* row_shr:2 */, 0xf, 0xf, true);
	v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x114 /* row_shr:4 */, 0xf, 0xf, true);
	v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x118 /* row_shr:8 */, 0xf, 0xf, true);
	SAU_DPP_ADD(0x142 /* row_bcast:15 */, 0xa);
	SAU_DPP_ADD(0x143 /* row_bcast:31 */, 0xc);
#undef SAU_DPP_ADD
	return v;
}

/* ... of 64-bit values (R oscillators' cycle | phase counters), mod 2^64. Round 6: two 32-bit scans -- each step one
 * v_add_u32 with the DPP move in it -- and the carries counted from one compare: the low words' inclusive sum P, taken mod 2^32,
 * wrapped on the way into lane i exactly when P[i] < lo[i] (sequentially P[i] = P[i-1] + lo[i]; the parallel scan ends with the same
 * P), and the carries into lane i's high word are the wraps of lanes 0..i: the bits of that compare's mask below and at the lane
 * (v_mbcnt_lo/hi + the lane's own bit). 16 vector instructions where 6 x (two DPP moves + a 64-bit add behind copies) took 32
 * per row of an R oscillator with a swept rate (profiles/census/r06_lookback_census.json). */
__device__ __forceinline__ unsigned long long wave_incl_scan64_dpp(unsigned long long v) {
	const uint32_t lo = (uint32_t)v;
	const uint32_t P = wave_incl_scan_dpp(lo);
	uint32_t H = wave_incl_scan_dpp((uint32_t)(v >> 32));
	const bool wrapped = P < lo;
	const unsigned long long m = __ballot(wrapped);
	H += __builtin_amdgcn_mbcnt_hi((uint32_t)(m >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m, 0u)) + (wrapped ? 1u : 0u);
	return ((unsigned long long)H << 32) | P;
}
/* the sum over all lanes of 64-bit values, mod 2^64 (look-back: only the total of the words looked at is wanted): the same
 * two scans' last lanes and the count of wraps */
__device__ __forceinline__ unsigned long long wave_sum64_dpp(unsigned long long v) {
	const uint32_t lo = (uint32_t)v;
	const uint32_t P = wave_incl_scan_dpp(lo);
	const uint32_t H = wave_incl_scan_dpp((uint32_t)(v >> 32));
	const unsigned long long m = __ballot(P < lo);
	const uint32_t tot_lo = (uint32_t)__builtin_amdgcn_readlane((int)P, 63);
	const uint32_t tot_hi = (uint32_t)__builtin_amdgcn_readlane((int)H, 63) + (uint32_t)__builtin_popcountll(m);
	return ((unsigned long long)tot_hi << 32) | tot_lo;
}
__device__ __forceinline__ unsigned long long readlane64(unsigned long long v, int lane) {
	const uint32_t lo = (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)v, lane);
	const uint32_t hi = (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)(v >> 32), lane);
	return ((unsigned long long)hi << 32) | lo;
}

#endif
