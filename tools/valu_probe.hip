/* valu_probe.hip -- what a wave64 vector instruction costs a gfx950 SIMD, per instruction class (round 5):
 *   hipcc --offload-arch=gfx950 -O3 tools/valu_probe.hip -o tools/valu_probe && tools/valu_probe [out.json]
 * The question: MI355X_MICROARCH.md rows 54 / 473 give 2 cycles per wave64 f32 instruction once a second wave shares the SIMD
 * ("SIMD-32"), 4 for one wave alone; bench.py's VALU fraction has priced EVERY vector instruction at 4. Which is it, for the
 * classes fast_kernel<12, 0, false, true> is made of (f64 add / mul, conversions, DPP moves, f32 / integer work, v_rcp)?
 * Method: one workgroup per CU (100 KB of LDS asked for, so no second workgroup joins), W waves per SIMD (workgroups of
 * 256 x W threads), every wave runs the same stream of INDEPENDENT instructions of one class over 16 register streams
 * (no instruction waits for the one before it: this is the issue rate, not the latency) and takes s_memtime before and
 * after; cycles per instruction per SIMD = wave's ticks / (instructions per wave x W). Wall time by hipEvent beside it gives
 * the clock: s_memtime counts shader cycles (MI355X_MICROARCH.md, constants table), s_memrealtime the fixed 100 MHz reference --
 * both are printed, so the ratio (the clock the chip really ran at under each stream) is measured, not assumed. The last rows are the hot path's own mix, with and without its LDS gathers. */
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
#include <string>
#include <vector>
#include <algorithm>

#define HIP_OK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

constexpr int NS = 16;       /* independent register streams */
constexpr int REPS = 8;      /* x NS instructions per loop turn */
constexpr int TURNS = 2048;   /* loop turns */

enum Cls { FMA32, ADD32, MUL32, PKFMA32, ADD64, MUL64, FMA64, CVT_F64_F32, CVT_F32_F64, CVT_F64_U32, CVT_I32_F64,
	DPP_SHR, DPP_ROR, DPP_ROWSHR, RCP32, ADDU32, LSHLADD, CNDMASK, AND32, MOV32, READLANE,
	CNDMASK_VCC1, CNDMASK_E64, CMP_E32, CMP_E64, FMAC32, SUB32, RNDNE32, CVT_I32_F32, CVT_F32_I32, LSHRREV, XOR32, MULLO, PKMUL32, PKADD32, MOV64, LSHLADD64,
	LDEXP64, FLOOR64, ADD32_E64, SUBU32_DPP, MAX3, WRITELANE, DSR128, DSR128_SAME, DSR64, DSW32,
	ALT_F64_U32, ALT_F64_DPP, ALT_DPP_U32, ALT_CVT_U32, ALT_F64_CNDVCC, MIX, MIX_LDS, MIX_NODPP, MIX_MOV, NCLS };
static const char *cls_name[NCLS] = {"v_fma_f32", "v_add_f32", "v_mul_f32", "v_pk_fma_f32", "v_add_f64", "v_mul_f64", "v_fma_f64",
	"v_cvt_f64_f32", "v_cvt_f32_f64", "v_cvt_f64_u32", "v_cvt_i32_f64", "v_mov_b32_dpp wave_shr:1", "v_mov_b32_dpp wave_ror:1",
	"v_mov_b32_dpp row_shr:1", "v_rcp_f32", "v_add_u32", "v_lshl_add_u32", "v_cndmask_b32", "v_and_b32", "v_mov_b32",
	"v_readlane_b32 (to SGPR)",
	"v_cndmask_b32_e32 (vcc set once before the loop)", "v_cndmask_b32_e64 (mask in an SGPR pair)", "v_cmp_eq_u32_e32 (to vcc)", "v_cmp_eq_u32_e64 (to an SGPR pair)",
	"v_fmac_f32_e32", "v_sub_f32_e32", "v_rndne_f32", "v_cvt_i32_f32", "v_cvt_f32_i32", "v_lshrrev_b32_e32", "v_xor_b32_e32", "v_mul_lo_u32", "v_pk_mul_f32", "v_pk_add_f32", "v_mov_b64", "v_lshl_add_u64",
	"v_ldexp_f64", "v_floor_f64", "v_add_f32_e64 (VOP3 encoding, |src| modifier)", "v_sub_u32_dpp wave_shr:1 (DPP on the VOP2 itself)", "v_max3_f32", "v_writelane_b32",
	"ds_read_b128 (hashed addresses, as the table gather)", "ds_read_b128 (lane-contiguous, conflict-free)", "ds_read_b64 (hashed addresses)", "ds_write_b32 (lane-contiguous)",
	"alternating v_add_f64 / v_add_u32 (per instruction)", "alternating v_add_f64 / v_mov_b32_dpp", "alternating v_mov_b32_dpp / v_add_u32", "alternating v_cvt_f64_u32 / v_add_u32", "alternating v_add_f64 / v_cndmask_b32_e32 vcc", "hot-path mix (16 f64-class + 10 DPP + 10 f32/int per 36)", "hot-path mix + 2 ds_read_b128 gathers + 1 ds_write_b32 per 36",
	"hot-path mix with its 10 DPP moves taken out (26 per block)", "hot-path mix with plain v_mov_b32 in place of the DPP moves"};

typedef uint32_t __attribute__((ext_vector_type(4))) u32x4;
template <int C>
__device__ __forceinline__ void body(float (&f)[NS], double (&d)[NS], uint32_t (&u)[NS], const float fa, const double da, const uint32_t lds_base, const unsigned long long smask, const uint32_t lane16) {
#pragma unroll
	for (int r = 0; r < REPS; ++r) {
#pragma unroll
		for (int i = 0; i < NS; ++i) {
			if (C == FMA32) asm volatile("v_fma_f32 %0, %0, %1, %1" : "+v"(f[i]) : "v"(fa));
			if (C == ADD32) asm volatile("v_add_f32 %0, %0, %1" : "+v"(f[i]) : "v"(fa));
			if (C == MUL32) asm volatile("v_mul_f32 %0, %0, %1" : "+v"(f[i]) : "v"(fa));
			if (C == PKFMA32) asm volatile("v_pk_fma_f32 %0, %0, %1, %1" : "+v"(d[i]) : "v"(da));
			if (C == ADD64) asm volatile("v_add_f64 %0, %0, %1" : "+v"(d[i]) : "v"(da));
			if (C == MUL64) asm volatile("v_mul_f64 %0, %0, %1" : "+v"(d[i]) : "v"(da));
			if (C == FMA64) asm volatile("v_fma_f64 %0, %0, %1, %1" : "+v"(d[i]) : "v"(da));
			if (C == CVT_F64_F32) asm volatile("v_cvt_f64_f32 %0, %1" : "=v"(d[i]) : "v"(f[i]));
			if (C == CVT_F32_F64) asm volatile("v_cvt_f32_f64 %0, %1" : "=v"(f[i]) : "v"(d[i]));
			if (C == CVT_F64_U32) asm volatile("v_cvt_f64_u32 %0, %1" : "=v"(d[i]) : "v"(u[i]));
			if (C == CVT_I32_F64) asm volatile("v_cvt_i32_f64 %0, %1" : "=v"(u[i]) : "v"(d[i]));
			if (C == DPP_SHR) asm volatile("v_mov_b32_dpp %0, %1 wave_shr:1 row_mask:0xf bank_mask:0xf" : "+v"(u[i]) : "v"(u[(i + 1) % NS]));
			if (C == DPP_ROR) asm volatile("v_mov_b32_dpp %0, %1 wave_ror:1 row_mask:0xf bank_mask:0xf bound_ctrl:0" : "+v"(u[i]) : "v"(u[(i + 1) % NS]));
			if (C == DPP_ROWSHR) asm volatile("v_mov_b32_dpp %0, %1 row_shr:1 row_mask:0xf bank_mask:0xf" : "+v"(u[i]) : "v"(u[(i + 1) % NS]));
			if (C == RCP32) asm volatile("v_rcp_f32 %0, %0" : "+v"(f[i]));
			if (C == ADDU32) asm volatile("v_add_u32 %0, %0, %1" : "+v"(u[i]) : "v"(u[(i + 1) % NS]));
			if (C == LSHLADD) asm volatile("v_lshl_add_u32 %0, %0, 1, %1" : "+v"(u[i]) : "v"(u[(i + 1) % NS]));
			if (C == CNDMASK) asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(u[i]) : "v"(u[(i + 1) % NS]));
			if (C == AND32) asm volatile("v_and_b32 %0, %0, %1" : "+v"(u[i]) : "v"(u[(i + 1) % NS]));
			if (C == MOV32) asm volatile("v_mov_b32 %0, %1" : "=v"(u[i]) : "v"(u[(i + 1) % NS]));
			if (C == CNDMASK_VCC1) asm volatile("v_cndmask_b32_e32 %0, %0, %1, vcc" : "+v"(u[i]) : "v"(u[(i + 1) % NS]));
			if (C == CNDMASK_E64) asm volatile("v_cndmask_b32_e64 %0, %0, %1, %2" : "+v"(u[i]) : "v"(u[(i + 1) % NS]), "s"(smask));
			if (C == CMP_E32) asm volatile("v_cmp_eq_u32_e32 vcc, %0, %1" :: "v"(u[i]), "v"(u[(i + 1) % NS]) : "vcc");
			if (C == CMP_E64) { unsigned long long m; asm volatile("v_cmp_eq_u32_e64 %0, %1, %2" : "=s"(m) : "v"(u[i]), "v"(u[(i + 1) % NS])); asm volatile("" :: "s"(m)); }
			if (C == FMAC32) asm volatile("v_fmac_f32_e32 %0, %1, %1" : "+v"(f[i]) : "v"(fa));
			if (C == SUB32) asm volatile("v_sub_f32_e32 %0, %0, %1" : "+v"(f[i]) : "v"(fa));
			if (C == RNDNE32) asm volatile("v_rndne_f32_e32 %0, %0" : "+v"(f[i]));
			if (C == CVT_I32_F32) asm volatile("v_cvt_i32_f32_e32 %0, %1" : "=v"(u[i]) : "v"(f[i]));
			if (C == CVT_F32_I32) asm volatile("v_cvt_f32_i32_e32 %0, %1" : "=v"(f[i]) : "v"(u[i]));
			if (C == LSHRREV) asm volatile("v_lshrrev_b32_e32 %0, 3, %1" : "=v"(u[i]) : "v"(u[(i + 1) % NS]));
			if (C == XOR32) asm volatile("v_xor_b32_e32 %0, %0, %1" : "+v"(u[i]) : "v"(u[(i + 1) % NS]));
			if (C == MULLO) asm volatile("v_mul_lo_u32 %0, %0, %1" : "+v"(u[i]) : "v"(u[(i + 1) % NS]));
			if (C == PKMUL32) asm volatile("v_pk_mul_f32 %0, %0, %1" : "+v"(d[i]) : "v"(da));
			if (C == PKADD32) asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(d[i]) : "v"(da));
			if (C == MOV64) asm volatile("v_mov_b64 %0, %1" : "=v"(d[i]) : "v"(d[(i + 1) % NS]));
			if (C == LSHLADD64) asm volatile("v_lshl_add_u64 %0, %0, 1, %1" : "+v"(d[i]) : "v"(d[(i + 1) % NS]));
			if (C == LDEXP64) asm volatile("v_ldexp_f64 %0, %0, %1" : "+v"(d[i]) : "v"(u[i]));
			if (C == FLOOR64) asm volatile("v_floor_f64_e32 %0, %0" : "+v"(d[i]));
			if (C == ADD32_E64) asm volatile("v_add_f32_e64 %0, |%0|, %1" : "+v"(f[i]) : "v"(fa));
			if (C == SUBU32_DPP) asm volatile("v_sub_u32_dpp %0, %1, %1 wave_shr:1 row_mask:0xf bank_mask:0xf" : "+v"(u[i]) : "v"(u[(i + 1) % NS]));
			if (C == MAX3) asm volatile("v_max3_f32 %0, %0, %1, %1" : "+v"(f[i]) : "v"(fa));
			if (C == WRITELANE) asm volatile("v_writelane_b32 %0, %1, 0" : "+v"(u[i]) : "s"((uint32_t)smask));
			if (C == DSR128) { u32x4 x; asm volatile("ds_read_b128 %0, %1" : "=v"(x) : "v"(lds_base + ((u[i] >> 17) & 0x7ff0))); asm volatile("" :: "v"(x)); }
			if (C == DSR128_SAME) { u32x4 x; asm volatile("ds_read_b128 %0, %1" : "=v"(x) : "v"(lds_base + (((u[i] >> 28) + (lds_base >> 31)) << 10) + lane16)); asm volatile("" :: "v"(x)); }
			if (C == DSR64) { double x; asm volatile("ds_read_b64 %0, %1" : "=v"(x) : "v"(lds_base + ((u[i] >> 18) & 0x3ff8))); asm volatile("" :: "v"(x)); }
			if (C == DSW32) asm volatile("ds_write_b32 %0, %1" :: "v"(lds_base + 65536u + lane16 / 4), "v"(f[i]));
			if (C == ALT_F64_U32) { if (i & 1) asm volatile("v_add_u32 %0, %0, %1" : "+v"(u[i]) : "v"(u[(i + 2) % NS])); else asm volatile("v_add_f64 %0, %0, %1" : "+v"(d[i]) : "v"(da)); }
			if (C == ALT_F64_DPP) { if (i & 1) asm volatile("v_mov_b32_dpp %0, %1 wave_shr:1 row_mask:0xf bank_mask:0xf" : "+v"(u[i]) : "v"(u[(i + 2) % NS])); else asm volatile("v_add_f64 %0, %0, %1" : "+v"(d[i]) : "v"(da)); }
			if (C == ALT_DPP_U32) { if (i & 1) asm volatile("v_add_u32 %0, %0, %1" : "+v"(u[i]) : "v"(u[(i + 2) % NS])); else asm volatile("v_mov_b32_dpp %0, %1 wave_shr:1 row_mask:0xf bank_mask:0xf" : "+v"(u[i]) : "v"(u[(i + 2) % NS])); }
			if (C == ALT_CVT_U32) { if (i & 1) asm volatile("v_add_u32 %0, %0, %1" : "+v"(u[i]) : "v"(u[(i + 2) % NS])); else asm volatile("v_cvt_f64_u32 %0, %1" : "=v"(d[i]) : "v"(u[i])); }
			if (C == ALT_F64_CNDVCC) { if (i & 1) asm volatile("v_cndmask_b32_e32 %0, %0, %1, vcc" : "+v"(u[i]) : "v"(u[(i + 2) % NS])); else asm volatile("v_add_f64 %0, %0, %1" : "+v"(d[i]) : "v"(da)); }
			if (C == READLANE) { uint32_t s; asm volatile("v_readlane_b32 %0, %1, 63" : "=s"(s) : "v"(u[i])); asm volatile("" :: "s"(s)); }
		}
	}
}

/* the hot path's own mix, per 36 vector instructions (VERDICT r04's census of fast_kernel<12, 0, false, true>'s PM path:
 * v_add_f64 7, v_mul_f64 5, conversions 4, v_mov_b32_dpp 10, other f32 / integer 10): nine blocks of four, no instruction
 * depends on one nearer than 16 back. LDS: two ds_read_b128 at hashed table addresses and a ds_write_b32, as a step has. */
template <bool LDS, int DPPMODE = 0>
__device__ __forceinline__ void mix_body(float (&f)[NS], double (&d)[NS], uint32_t (&u)[NS], const uint32_t (&w)[4], const float fa, const double da, const uint32_t lds_base, const uint32_t lane) {
#pragma unroll
	for (int r = 0; r < REPS * NS / 36 + 1; ++r) {
		const int a = (r * 5) % NS, b = (r * 5 + 1) % NS, c = (r * 5 + 2) % NS, e = (r * 5 + 3) % NS, g = (r * 5 + 4) % NS;
		if (LDS) {
			u32x4 x, y;
			asm volatile("ds_read_b128 %0, %1" : "=v"(x) : "v"(lds_base + ((u[a] >> 17) & 0x7ff0)));
			asm volatile("ds_read_b128 %0, %1 offset:32768" : "=v"(y) : "v"(lds_base + ((u[b] >> 17) & 0x7ff0)));
			asm volatile("ds_write_b32 %0, %1" :: "v"(lds_base + 65536u + lane * 4), "v"(f[c]));
			asm volatile("s_waitcnt lgkmcnt(2)" ::: "memory");
			asm volatile("" :: "v"(x), "v"(y));
		}
		asm volatile("v_add_f64 %0, %0, %1" : "+v"(d[a]) : "v"(da));
		if (DPPMODE == 0) asm volatile("v_mov_b32_dpp %0, %1 wave_shr:1 row_mask:0xf bank_mask:0xf" : "+v"(u[a]) : "v"(w[1]));
		if (DPPMODE == 2) asm volatile("v_mov_b32 %0, %1" : "=v"(u[a]) : "v"(w[1]));
		asm volatile("v_mul_f64 %0, %0, %1" : "+v"(d[b]) : "v"(da));
		asm volatile("v_add_u32 %0, %0, %1" : "+v"(u[c]) : "v"(u[e]));
		asm volatile("v_add_f64 %0, %0, %1" : "+v"(d[c]) : "v"(da));
		if (DPPMODE == 0) asm volatile("v_mov_b32_dpp %0, %1 wave_ror:1 row_mask:0xf bank_mask:0xf bound_ctrl:0" : "+v"(u[b]) : "v"(w[2]));
		if (DPPMODE == 2) asm volatile("v_mov_b32 %0, %1" : "=v"(u[b]) : "v"(w[2]));
		asm volatile("v_cvt_f64_u32 %0, %1" : "=v"(d[e]) : "v"(u[g]));
		asm volatile("v_lshrrev_b32 %0, 21, %1" : "=v"(u[e]) : "v"(u[a]));
		asm volatile("v_mul_f64 %0, %0, %1" : "+v"(d[g]) : "v"(da));
		if (DPPMODE == 0) asm volatile("v_mov_b32_dpp %0, %1 wave_shr:1 row_mask:0xf bank_mask:0xf" : "+v"(u[g]) : "v"(w[0]));
		if (DPPMODE == 2) asm volatile("v_mov_b32 %0, %1" : "=v"(u[g]) : "v"(w[0]));
		asm volatile("v_add_f64 %0, %0, %1" : "+v"(d[a]) : "v"(da));
		asm volatile("v_and_b32 %0, 0x1fffff, %1" : "=v"(u[c]) : "v"(u[b]));
		if (DPPMODE == 0) asm volatile("v_mov_b32_dpp %0, %1 wave_ror:1 row_mask:0xf bank_mask:0xf bound_ctrl:0" : "+v"(u[e]) : "v"(w[0]));
		if (DPPMODE == 2) asm volatile("v_mov_b32 %0, %1" : "=v"(u[e]) : "v"(w[0]));
		asm volatile("v_mul_f64 %0, %0, %1" : "+v"(d[b]) : "v"(da));
		asm volatile("v_sub_u32 %0, %0, %1" : "+v"(u[a]) : "v"(u[c]));
		asm volatile("v_add_f64 %0, %0, %1" : "+v"(d[c]) : "v"(da));
		if (DPPMODE == 0) asm volatile("v_mov_b32_dpp %0, %1 wave_shr:1 row_mask:0xf bank_mask:0xf" : "+v"(u[c]) : "v"(w[3]));
		if (DPPMODE == 2) asm volatile("v_mov_b32 %0, %1" : "=v"(u[c]) : "v"(w[3]));
		asm volatile("v_cvt_f32_f64 %0, %1" : "=v"(f[a]) : "v"(d[e]));
		asm volatile("v_rcp_f32 %0, %1" : "=v"(f[b]) : "v"(f[c]));
		if (DPPMODE == 0) asm volatile("v_mov_b32_dpp %0, %1 wave_ror:1 row_mask:0xf bank_mask:0xf bound_ctrl:0" : "+v"(u[g]) : "v"(w[0]));
		if (DPPMODE == 2) asm volatile("v_mov_b32 %0, %1" : "=v"(u[g]) : "v"(w[0]));
		asm volatile("v_add_f64 %0, %0, %1" : "+v"(d[g]) : "v"(da));
		asm volatile("v_fma_f32 %0, %0, %1, %1" : "+v"(f[e]) : "v"(fa));
		asm volatile("v_mul_f64 %0, %0, %1" : "+v"(d[a]) : "v"(da));
		if (DPPMODE == 0) asm volatile("v_mov_b32_dpp %0, %1 wave_shr:1 row_mask:0xf bank_mask:0xf" : "+v"(u[b]) : "v"(w[0]));
		if (DPPMODE == 2) asm volatile("v_mov_b32 %0, %1" : "=v"(u[b]) : "v"(w[0]));
		asm volatile("v_cvt_f64_f32 %0, %1" : "=v"(d[b]) : "v"(f[g]));
		asm volatile("v_mul_f32 %0, %0, %1" : "+v"(f[c]) : "v"(fa));
		asm volatile("v_add_f64 %0, %0, %1" : "+v"(d[c]) : "v"(da));
		if (DPPMODE == 0) asm volatile("v_mov_b32_dpp %0, %1 wave_ror:1 row_mask:0xf bank_mask:0xf bound_ctrl:0" : "+v"(u[a]) : "v"(w[1]));
		if (DPPMODE == 2) asm volatile("v_mov_b32 %0, %1" : "=v"(u[a]) : "v"(w[1]));
		asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(u[e]) : "v"(u[c]));
		asm volatile("v_mul_f64 %0, %0, %1" : "+v"(d[e]) : "v"(da));
		if (DPPMODE == 0) asm volatile("v_mov_b32_dpp %0, %1 wave_shr:1 row_mask:0xf bank_mask:0xf" : "+v"(u[c]) : "v"(w[0]));
		if (DPPMODE == 2) asm volatile("v_mov_b32 %0, %1" : "=v"(u[c]) : "v"(w[0]));
		asm volatile("v_cvt_f64_f32 %0, %1" : "=v"(d[g]) : "v"(f[a]));
		asm volatile("v_cmp_eq_u32 vcc, %0, %1" :: "v"(u[b]), "v"(u[e]) : "vcc");
		asm volatile("v_add_f64 %0, %0, %1" : "+v"(d[a]) : "v"(da));
		if (DPPMODE == 0) asm volatile("v_mov_b32_dpp %0, %1 wave_ror:1 row_mask:0xf bank_mask:0xf bound_ctrl:0" : "+v"(u[e]) : "v"(w[0]));
		if (DPPMODE == 2) asm volatile("v_mov_b32 %0, %1" : "=v"(u[e]) : "v"(w[0]));
		asm volatile("v_fma_f32 %0, %0, %1, %1" : "+v"(f[g]) : "v"(fa));
	}
}
constexpr int MIX_PER_TURN = (REPS * NS / 36 + 1) * 36;
constexpr int MIX_LDS_ADDR_PER_TURN = (REPS * NS / 36 + 1) * 6; /* v_lshrrev + v_and + v_add per gather address: vector instructions too */

template <int C>
__global__ void __launch_bounds__(1024) probe(unsigned long long *ticks, float *sink, const float fa, const double da) {
	extern __shared__ __align__(16) unsigned char lds[];
	const uint32_t lane = threadIdx.x & 63;
	float f[NS]; double d[NS]; uint32_t u[NS];
#pragma unroll
	for (int i = 0; i < NS; ++i) { f[i] = 1.0f + 0.001f * (float)(i + lane); d[i] = 1.0 + 1e-6 * (double)(i + lane); u[i] = 0x9e3779b9u * (lane * NS + i + 1); }
	const uint32_t lds_base = (uint32_t)(uintptr_t)lds;
	uint32_t w[4] = {u[0] ^ 1u, u[1] ^ 2u, u[2] ^ 3u, u[3] ^ 4u};
#pragma unroll
	for (int i = 0; i < 4; ++i) asm volatile("" : "+v"(w[i]));
	unsigned long long smask = 0x5555555555555555ull ^ (unsigned long long)lds_base;
	asm volatile("s_mov_b64 vcc, %0" :: "s"(smask) : "vcc");
	if (C == MIX_LDS || C == DSR128 || C == DSR128_SAME || C == DSR64 || C == DSW32) { for (uint32_t i = threadIdx.x; i < 98304 / 4; i += blockDim.x) ((uint32_t *)lds)[i] = i; }
	__syncthreads();
	const unsigned long long t0 = __builtin_amdgcn_s_memtime();
	const unsigned long long c0 = __builtin_amdgcn_s_memrealtime();
	for (int t = 0; t < TURNS; ++t) {
		if (C == DSR128 || C == DSR128_SAME || C == DSR64 || C == DSW32) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
		if (C == MIX) mix_body<false>(f, d, u, w, fa, da, lds_base, lane);
		else if (C == MIX_LDS) mix_body<true>(f, d, u, w, fa, da, lds_base, lane);
		else if (C == MIX_NODPP) mix_body<false, 1>(f, d, u, w, fa, da, lds_base, lane);
		else if (C == MIX_MOV) mix_body<false, 2>(f, d, u, w, fa, da, lds_base, lane);
		else body<C>(f, d, u, fa, da, lds_base, smask, lane * 16);
	}
	asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
	const unsigned long long c1 = __builtin_amdgcn_s_memrealtime();
	const unsigned long long t1 = __builtin_amdgcn_s_memtime();
	float acc = 0;
#pragma unroll
	for (int i = 0; i < NS; ++i) acc += f[i] + (float)d[i] + (float)u[i];
	if (acc == 12345.678f) sink[threadIdx.x] = acc;
	if (lane == 0) {
		const uint32_t w = (blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6)) * 2;
		ticks[w] = t1 - t0; ticks[w + 1] = c1 - c0;
	}
}

struct Row { std::string name; int waves_per_simd; double memtime_ticks, cyclecounter_ticks, wall_ms, insts_per_wave; };

template <int C>
static Row run(int wps) {
	const int threads = 256 * wps, blocks = 256;
	const size_t lds = 100 * 1024;
	HIP_OK(hipFuncSetAttribute((const void *)probe<C>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
	unsigned long long *ticks; float *sink;
	const size_t nw = (size_t)blocks * (threads / 64);
	HIP_OK(hipMalloc(&ticks, nw * 16)); HIP_OK(hipMalloc(&sink, 4096));
	hipEvent_t a, b; HIP_OK(hipEventCreate(&a)); HIP_OK(hipEventCreate(&b));
	float best = 1e9f; std::vector<unsigned long long> h(nw * 2), hb;
	for (int rep = 0; rep < 4; ++rep) {
		HIP_OK(hipEventRecord(a));
		hipLaunchKernelGGL((probe<C>), dim3(blocks), dim3(threads), lds, 0, ticks, sink, 1.0000001f, 1.0000000001);
		HIP_OK(hipEventRecord(b)); HIP_OK(hipEventSynchronize(b));
		float ms; HIP_OK(hipEventElapsedTime(&ms, a, b));
		if (rep && ms < best) { best = ms; HIP_OK(hipMemcpy(h.data(), ticks, nw * 16, hipMemcpyDeviceToHost)); hb = h; }
	}
	double mt = 0, cc = 0;
	for (size_t i = 0; i < nw; ++i) { mt += (double)hb[2 * i]; cc += (double)hb[2 * i + 1]; }
	HIP_OK(hipFree(ticks)); HIP_OK(hipFree(sink));
	const double ipw = (C == MIX || C == MIX_MOV) ? (double)TURNS * MIX_PER_TURN : C == MIX_NODPP ? (double)TURNS * MIX_PER_TURN * 26 / 36 : C == MIX_LDS ? (double)TURNS * (MIX_PER_TURN + MIX_LDS_ADDR_PER_TURN) : (double)TURNS * REPS * NS;
	return Row{cls_name[C], wps, mt / nw, cc / nw, best, ipw};
}

int main(int argc, char **argv) {
	hipDeviceProp_t p; HIP_OK(hipGetDeviceProperties(&p, 0));
	int wall_khz = 0; HIP_OK(hipDeviceGetAttribute(&wall_khz, hipDeviceAttributeWallClockRate, 0));
	std::vector<Row> rows;
	for (int wps : {1, 2, 4}) {
#define R(C) rows.push_back(run<C>(wps));
		R(FMA32) R(ADD32) R(MUL32) R(PKFMA32) R(ADD64) R(MUL64) R(FMA64) R(CVT_F64_F32) R(CVT_F32_F64) R(CVT_F64_U32) R(CVT_I32_F64)
		R(DPP_SHR) R(DPP_ROR) R(DPP_ROWSHR) R(RCP32) R(ADDU32) R(LSHLADD) R(CNDMASK) R(AND32) R(MOV32) R(READLANE)
		R(CNDMASK_VCC1) R(CNDMASK_E64) R(CMP_E32) R(CMP_E64) R(FMAC32) R(SUB32) R(RNDNE32) R(CVT_I32_F32) R(CVT_F32_I32) R(LSHRREV) R(XOR32) R(MULLO) R(PKMUL32) R(PKADD32) R(MOV64) R(LSHLADD64)
		R(LDEXP64) R(FLOOR64) R(ADD32_E64) R(SUBU32_DPP) R(MAX3) R(WRITELANE) R(DSR128) R(DSR128_SAME) R(DSR64) R(DSW32)
		R(ALT_F64_U32) R(ALT_F64_DPP) R(ALT_DPP_U32) R(ALT_CVT_U32) R(ALT_F64_CNDVCC) R(MIX) R(MIX_LDS) R(MIX_NODPP) R(MIX_MOV)
#undef R
	}
	/* the shader clock during a run: instructions are counted, so cycles = wall time x f; f from the v_add_f64 row (a plain
	 * full-rate f64 add is 4 cycles on every CDNA part: 16 lanes per cycle, the 78.6 TF vector f64 peak = 256 CUs x 4 SIMDs
	 * x 16 lanes x 2 flop x 2.4 GHz) is NOT assumed: s_memtime's rate is measured against the wall clock below instead. */
	FILE *o = argc > 1 ? fopen(argv[1], "w") : stdout;
	fprintf(o, "{\"device\": \"%s\", \"cus\": %d, \"clock_khz_reported\": %d, \"wall_clock_khz\": %d, \"insts\": \"independent, %d register streams\",\n \"rows\": [\n",
		p.gcnArchName, p.multiProcessorCount, p.clockRate, wall_khz, NS);
	for (size_t i = 0; i < rows.size(); ++i) {
		const Row &r = rows[i];
		/* kernel wall time ~ wave time (all waves start together, one workgroup per CU): memtime tick rate = ticks / wall */
		const double tick_hz = r.memtime_ticks / (r.wall_ms * 1e-3);
		const double ns_per_inst_simd = r.wall_ms * 1e6 / (r.insts_per_wave * r.waves_per_simd);
		fprintf(o, "  {\"inst\": \"%s\", \"waves_per_simd\": %d, \"insts_per_wave\": %.0f, \"wall_ms\": %.4f, \"memtime_ticks_per_wave\": %.0f, "
			"\"memrealtime_ticks_per_wave\": %.0f, \"memtime_tick_hz_if_wave_spans_launch\": %.4g, \"ns_per_inst_per_simd\": %.4f, "
			"\"clock_ghz_measured\": %.4f, \"cycles_per_inst_per_simd\": %.3f, \"avg_wave_ticks_per_inst_x_waves\": %.4f}%s\n",
			r.name.c_str(), r.waves_per_simd, r.insts_per_wave, r.wall_ms, r.memtime_ticks, r.cyclecounter_ticks, tick_hz, ns_per_inst_simd,
			r.memtime_ticks / r.cyclecounter_ticks * 0.1, ns_per_inst_simd * r.memtime_ticks / r.cyclecounter_ticks * 0.1,
			r.memtime_ticks / (r.insts_per_wave * r.waves_per_simd), i + 1 < rows.size() ? "," : "");
	}
	fprintf(o, " ]}\n");
	if (o != stdout) fclose(o);
	/* a table for people */
	printf("%-72s %5s %10s %9s %12s\n", "instruction (independent stream)", "w/SIMD", "ns/inst/SIMD", "clock GHz", "cycles/inst/SIMD");
	for (const Row &r : rows) {
		const double ns = r.wall_ms * 1e6 / (r.insts_per_wave * r.waves_per_simd), ghz = r.memtime_ticks / r.cyclecounter_ticks * 0.1;
		printf("%-72s %5d %10.4f %9.3f %12.3f\n", r.name.c_str(), r.waves_per_simd, ns, ghz, ns * ghz);
	}
	return 0;
}
