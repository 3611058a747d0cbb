/* sau_dev_math.h -- per-sample arithmetic of the generator hot path, written
 * once for both sides of the PCIe bus: the HIP kernels (kernels.hip) include
 * it as device code, host code (event application mirror, table builder) and
 * the test-side sequential plan executor include it as plain C++.
 *
 * Every function states the reference expression it reproduces (saugns
 * v0.4.7 file:line).  Evaluation order is exactly the order written here:
 * build with -ffp-contract=off and without fast-math.  Where the reference's
 * own build (gcc -O3 -ffast-math on sau/line.c and sau/generator.c)
 * evaluates in a different association than its C source, the association of
 * that build's main loops is used (marked "ref-build form"); see DESIGN.md
 * "Arithmetic contract".
 */
#ifndef SAU_DEV_MATH_H
#define SAU_DEV_MATH_H

#include <stdint.h>
#include <math.h>

#if defined(__HIPCC__)
#include <hip/hip_runtime.h>
#define SAU_HD __host__ __device__ __forceinline__
/* larger helpers: real functions on the device (smaller kernels, and the line
 * code is then compiled once instead of once per call site) */
#define SAU_HD_CALL __host__ __device__ inline __attribute__((noinline))
#else
#define SAU_HD static inline
#define SAU_HD_CALL static inline
#endif

namespace saudev {

/* ---- constants of the data contract (include/sau_abi.h) ----------------- */
enum : uint32_t {
	LP_STATE = 1, LP_STATE_RATIO = 2, LP_GOAL = 4, LP_GOAL_RATIO = 8,
	LP_TYPE = 16, LP_TIME = 32, LP_TIME_IF_NEW = 64,
};
enum : uint32_t {
	LN_cos = 0, LN_lin, LN_sah, LN_exp, LN_log, LN_xpe, LN_lge, LN_sqe,
	LN_cub, LN_smo, LN_ncl, LN_nhl, LN_uwh, LN_COUNT
};
enum : uint32_t { OT_AMP = 0, OT_NOISE, OT_WAVE, OT_RASEG };
enum : uint32_t { NZ_wh = 0, NZ_gw, NZ_bw, NZ_tw, NZ_re, NZ_vi, NZ_bv };
enum : uint32_t {
	RF_URAND = 0, RF_GAUSS, RF_BIN, RF_TERN, RF_FIXED, RF_ADDREC
};
enum : uint32_t {
	RO_PERLIN = 1, RO_HALFSHAPE = 2, RO_ZIGZAG = 4, RO_SQUARE = 8, RO_VIOLET = 16,
	RO_LINE_SET = 1 << 6, RO_FUNC_SET = 1 << 7, RO_LEVEL_SET = 1 << 8,
	RO_ASUBVAL_SET = 1 << 9,
};

constexpr uint32_t WAVE_LEN = 2048;
constexpr uint32_t WAVE_MASK = WAVE_LEN - 1;
constexpr uint32_t SLEN_BITS = 21;           /* sau/wave.h:27 */
constexpr uint32_t SLEN = 1u << SLEN_BITS;

/* ---- integer helpers ----------------------------------------------------- */

/* sau/math.h:63-64 + generator.c:17: llrintf, kept as int64 */
SAU_HD int64_t rint64(float x) {
#if defined(__HIP_DEVICE_COMPILE__)
	/* out of range (and NaN) the reference's host gives the x86 "integer indefinite", 0x8000...0 (cvtss2si), where
	 * the device's conversion saturates: frequency-scaled PM of thousands of cycles under a frequency of gigahertz
	 * gets there (found by the sweep with extreme parameters, round 3) */
	/* (gfx950 has no f32 -> i64 conversion: the compiler's costs eleven vector instructions. Phase increments are almost always
	 * below 2^31 in magnitude -- a frequency below the sample rate -- and there v_rndne_f32 + v_cvt_i32_f32 + a sign extension do:
	 * a float of that size rounds to an integer an int32 holds. One wave-uniform test picks the form.) */
	if (__all(fabsf(x) < 0x1p31f)) return (long long)(int32_t)rintf(x);
	const long long v = __float2ll_rn(x);
	return fabsf(x) < 0x1p63f ? v : (long long)0x8000000000000000ull;
#else
	return llrintf(x);
#endif
}
/* (int32_t)x as the reference's host computes it: out of range (and NaN) cvttss2si gives 0x80000000, the device saturates */
SAU_HD int32_t f2i_x86(float x) {
#if defined(__HIP_DEVICE_COMPILE__)
	return fabsf(x) < 0x1p31f ? (int32_t)x : (int32_t)0x80000000u;
#else
	return (int32_t)x;
#endif
}
/* floorf(x) taken to int32_t as the reference build's six sauRasG_map_*_s loops have it (rasg.h:251; -ffast-math
 * inlines floorf): truncate (cvttss2si), and where the truncated value lies above x take one off -- in 32-bit
 * integer arithmetic, so that below -2^31 the "integer indefinite" 0x80000000 wraps to INT32_MAX and the phase
 * moves the other way (oracle/_ref/generator.o; found by the batch sweep with extreme parameters, round 3) */
SAU_HD int32_t floor_i32_ref(float x) {
	int32_t t = f2i_x86(x);
	if ((float)t > x) t = (int32_t)((uint32_t)t - 1u);
	return t;
}
/* the feedback amount of the same loops: the build halves the amount first, fb_s * (0.5f * pm_a) -- the source's
 * (fb_s * pm_a) * 0.5f unless a product leaves the normal range */
SAU_HD float ras_fb_amount(float fb_s, float pm_a) { return fb_s * (0.5f * pm_a); }
/* ... and wrapped into a 32-bit phase */
SAU_HD uint32_t rint32w(float x) {
#if defined(__HIP_DEVICE_COMPILE__)
	/* gfx950 has no f32->i64 conversion; only the low 32 bits are needed.
	 * Small magnitudes convert directly; larger ones are reduced mod 2^32 in
	 * f64, where every step is exact (integer-valued operands < 2^128). */
	float r = rintf(x);
	if (fabsf(r) < 0x1p31f) return (uint32_t)(int32_t)r;
	double d = (double)r;
	double q = floor(d * 0x1p-32);
	return (uint32_t)(d - q * 0x1p32);
#else
	return (uint32_t)rint64(x);
#endif
}

/* rint(p * 2^31) wrapped to 32 bits (the PM offset of wosc.h:152-166).
 * Device form without a 64-bit conversion and without branches: with
 * r = rint(p/2), m = p - 2r is exact and |m| <= 1, and
 * p*2^31 = m*2^31 + r*2^32, so the wrapped results agree. */
SAU_HD uint32_t rint32w_p31(float p) {
#if defined(__HIP_DEVICE_COMPILE__)
	float r = rintf(p * 0.5f);
	float m = p - (r + r);
	float y = rintf(m * 0x1p31f);          /* in [-2^31, 2^31] */
	if (y >= 0x1p31f) y -= 0x1p32f;        /* wrap +2^31 to -2^31: now in int32 range */
	return (uint32_t)(int32_t)y;
#else
	return (uint32_t)rint64(p * 0x1p31f);
#endif
}

/* sau/math.h:297-303 */
SAU_HD uint32_t ranfast32(uint32_t n) {
	uint32_t s = n * 0x9e3779b9u;
	s ^= s >> 14;
	s = (s | 1u) * s;
	s ^= s >> 13;
	return s;
}
/* sau/math.h:283-285 */
SAU_HD uint32_t mcg32(uint32_t seed) { return seed * 0xe47135u; }
/* sau/math.h:94-96 */
/* (the shift count taken mod 32, as the host's scalar `sar` and the device's shifts take it: the reference's parser gives R
 * levels 0..30, sau/program.h:145-149 -- only a hand-made program image can ask for more, and there the reference's own
 * `x >> s` is undefined; found by the sanitizer fuzz of round 4) */
SAU_HD int32_t sar32(int32_t x, int s) { s &= 31; return x < 0 ? ~(~x >> s) : x >> s; }
/* sau/math.h:112-118 */
SAU_HD int32_t foldhd32(int32_t x) {
	uint32_t s = (uint32_t)x;
	if (s + (1u << 29) > (1u << 31))
		s = (1u << 31) + (1u << 30) - s;
	s = (s - (1u << 29)) * 2u;
	return (int32_t)s;
}
SAU_HD int odd_sign(uint32_t n) { return 1 - (int)((n & 1u) * 2u); } /* math.h:89 */
SAU_HD float fscalei(uint32_t i, float sc) { return (float)(int32_t)i * sc; } /* generator.c:19 */
SAU_HD int32_t divi(uint32_t i, int32_t d) { return ((int32_t)i) / d; }       /* generator.c:20 */
SAU_HD float bits_f(uint32_t u) { union { uint32_t u; float f; } c; c.u = u; return c.f; }
SAU_HD uint32_t f_bits(float f) { union { uint32_t u; float f; } c; c.f = f; return c.u; }

/* sau/math.h:366-379 */
SAU_HD float sinpi_d5f(float x) {
	const float k0 = +3.14042741234069229463f;
	const float k1 = -5.13655757476162831091f;
	const float k2 = +2.29939170159543653372f;
	float x2 = x * x;
	return x * (k0 + x2 * (k1 + x2 * k2));
}

/* ---- ramp shapes --------------------------------------------------------- */

/* sau/line.h:174-183 */
SAU_HD float sinramp(float x) {
	const float k0 = +1.5702137061703461473139223358864f;
	const float k1 = -2.568278787380814155456160152724f;
	const float k2 = +1.1496958507977182668618673644367f;
	float x2 = x * x;
	return x * (k0 + x2 * (k1 + x2 * k2));
}
/* sau/line.h:195-200, ref-build form: x3 + ((x*c1 + x2*c2)*(x3 - 1))*x2 */
SAU_HD float expramp6(float x) {
	float x2 = x * x;
	float x3 = x2 * x;
	return x3 + ((x * (629.f / 1792.f) + x2 * (1163.f / 1792.f)) * (x3 + -1.f)) * x2;
}

/* Scalar shape value, sau/line.h:153-266 (ref-build forms for cub smo ncl nhl
 * and, through expramp6, exp log xpe lge). */
SAU_HD float shape_val_inl(uint32_t type, float x, float a, float b) {
	switch (type) {
	default:
	case LN_sah: return a;
	case LN_lin: return a + (b - a) * x;
	case LN_cos: return a + (b - a) * (sinramp(x - 0.5f) + 0.5f);
	case LN_exp:
		return (a > b) ? b + (a - b) * expramp6(1.f - x) : a + (b - a) * expramp6(x);
	case LN_log:
		return (a < b) ? b + (a - b) * expramp6(1.f - x) : a + (b - a) * expramp6(x);
	case LN_xpe: return b + (a - b) * expramp6(1.f - x);
	case LN_lge: return a + (b - a) * expramp6(x);
	case LN_sqe: { float y = 1.f - x; return b + (a - b) * (y * y); }
	case LN_cub: {
		float y = (0.5f - x) * 2;
		return b + (y * y * y + 1.f) * ((a - b) * 0.5f);
	}
	case LN_smo:
		return a + (((b - a) * x) * (x * x)) * ((x * 6.f + -15.f) * x + 10.f);
	case LN_uwh: {
		int32_t s = (int32_t)ranfast32(f_bits(x));
		return a + (b - a) * (0.5f + (0.5f * 0x1p-31f) * (float)s);
	}
	case LN_ncl: {
		int32_t s = (int32_t)ranfast32(f_bits(x));
		float t = ((x + x) + -3.f) * x + 1.f;
		return a + (b - a) * ((t * (float)s) * (x * (0.5f * 0x1p-31f)) + x);
	}
	case LN_nhl: {
		int32_t s = (int32_t)ranfast32(f_bits(x));
		return a + (b - a) * (((float)s * (1.f - x)) * (x * 0x1p-31f) + x);
	}
	}
}
/* (a real call on the device: inlined into the 1000-line block-loop kernel it exposed a hipcc miscompile, DESIGN.md 5;
 * rchain_kernel -- a small kernel with the shape known per wave -- takes the inline form, ras_sample<true>) */
SAU_HD_CALL float shape_val(uint32_t type, float x, float a, float b) { return shape_val_inl(type, x, a, b); }

/* Parameters of one timed sweep, hoisted out of the per-sample evaluation
 * exactly as the reference's fill functions hoist them (sau/line.c:65-281). */
struct Sweep {
	uint32_t type;   /* LN_* with exp/log already resolved to xpe/lge */
	float v0, vt;
	uint32_t pos;    /* index of sample 0 of this block inside the sweep */
	int32_t adj_pos; /* pos - time/2 */
	float inv_time;  /* 1.f / time */
	float k;         /* shape-specific hoisted factor */
	float vm, vd;
};

SAU_HD_CALL Sweep sweep_setup(uint32_t type, float v0, float vt, uint32_t pos, uint32_t time) {
	Sweep s;
	if (type == LN_exp) type = (v0 > vt) ? LN_xpe : LN_lge; /* sau/line.c:125-131 */
	else if (type == LN_log) type = (v0 < vt) ? LN_xpe : LN_lge; /* 142-148 */
	s.type = type; s.v0 = v0; s.vt = vt; s.pos = pos;
	s.adj_pos = (int32_t)(pos - (time / 2));
	s.inv_time = 1.f / (float)time;
	s.vm = (v0 + vt) * 0.5f;
	s.vd = (vt - v0);
	s.k = 0.f;
	switch (type) {
	case LN_lin: s.k = s.vd * s.inv_time; break;          /* ref-build hoist */
	case LN_cub: s.k = -2 * s.inv_time; break;            /* sau/line.c:206 */
	case LN_uwh: s.vd = (vt - v0) * (0.5f / (float)INT32_MAX); break; /* :230 */
	default: break;
	}
	return s;
}

/* Value of sample i of the block (i + pos inside the sweep), before any
 * ratio multiplication. sau/line.c:27-37,65-281 in ref-build forms. */
template <bool INL = false>
SAU_HD float sweep_value_inl(const Sweep &s, uint32_t i) {
	switch (s.type) {
	default:
	case LN_sah: return s.v0;
	case LN_lin: return s.vm + s.k * (float)((int32_t)i + s.adj_pos);
	case LN_cos: {
		float x = (float)((int32_t)i + s.adj_pos) * s.inv_time;
		float x2 = x * x;
		return s.vm + (s.vd * x) * ((x2 * +1.1496958507977182668618673644367f
			+ -2.568278787380814155456160152724f) * x2
			+ +1.5702137061703461473139223358864f);
	}
	case LN_xpe: case LN_lge: case LN_smo: {
		float x = (float)(i + s.pos) * s.inv_time;
		if (INL) { /* the three shapes in place (the time-parallel kernels: a real call per value costs more than the value) */
			if (s.type == LN_xpe) return s.vt + (s.v0 - s.vt) * expramp6(1.f - x);
			if (s.type == LN_lge) return s.v0 + (s.vt - s.v0) * expramp6(x);
			return s.v0 + (((s.vt - s.v0) * x) * (x * x)) * ((x * 6.f + -15.f) * x + 10.f);
		}
		return shape_val(s.type, x, s.v0, s.vt);
	}
	case LN_sqe: {
		float x = 0.5f - (float)((int32_t)i + s.adj_pos) * s.inv_time;
		return s.vt + (s.v0 - s.vt) * (x * x);
	}
	case LN_cub: {
		float x = (float)((int32_t)i + s.adj_pos) * s.k;
		return s.vt + (x * x * x + 1.f) * ((s.v0 - s.vt) * 0.5f);
	}
	case LN_uwh: {
		int32_t r = (int32_t)ranfast32(s.pos + i);
		return s.vm + s.vd * (float)r;
	}
	case LN_ncl: {
		float x = (float)((int32_t)i + s.adj_pos) * s.inv_time;
		int32_t r = (int32_t)ranfast32(s.pos + i);
		float xb0 = x + 0.5f;
		float t = ((xb0 + xb0) + -3.f) * xb0 + 1.f;
		return s.vm + s.vd * (((float)r * t) * (xb0 * (0.5f / (float)INT32_MAX)) + x);
	}
	case LN_nhl: {
		float x = (float)((int32_t)i + s.adj_pos) * s.inv_time;
		int32_t r = (int32_t)ranfast32(s.pos + i);
		float xb0 = x + 0.5f;
		return s.vm + s.vd * (((float)r * (1.f - xb0)) * (xb0 * (2 * 0.5f / (float)INT32_MAX)) + x);
	}
	}
}

SAU_HD_CALL float sweep_value(const Sweep &s, uint32_t i) { return sweep_value_inl(s, i); }

/* The reference build's loop tails. gcc vectorises sauLine_fill_cub two samples at a time and sauLine_map_cub four at a
 * time, and gives the scalar epilogues another association than the vector bodies: x*x*x*h + h instead of
 * (x*x*x + 1)*h, h = (v0 - vt)/2 -- the last sample of a fill of odd length, the last len % 4 samples of a map. Every
 * other loop of sau/line.c has the same form in body and tail (the oracle's mode 2 = mode 1 + exactly these two,
 * and equals the compiled reference on 15000 random programs). `len` is the length of the reference's call: its
 * block -- 1024 frames from the start of its span, the span's last one shorter (Lattice below) -- cut where the
 * operator, one of its ancestors or the voice stops (generator.c:694-698, 833-846), and for a fill where the sweep
 * ends (sau/line.c:349-378). */
SAU_HD float sweep_cub_tail(const Sweep &s, uint32_t i) {
	const float x = (float)((int32_t)i + s.adj_pos) * s.k;
	const float h = (s.v0 - s.vt) * 0.5f;
	return s.vt + (x * x * x * h + h);
}
SAU_HD float shape_cub_tail(float x, float a, float b) {
	const float y = (0.5f - x) * 2;
	const float h = (a - b) * 0.5f;
	return b + (y * y * y * h + h);
}

/* ---- ramp state machine -------------------------------------------------- */

struct LineState { /* device copy of sauLine minus time_ms */
	float v0, vt;
	uint32_t pos, end;
	uint32_t type;  /* LN_* */
	uint32_t flags; /* LP_* in the low byte; lattice carry above (LPX_*) */
};

/* The reference renders in blocks of at most 1024 frames that start anew at every
 * sauGenerator_run call and at every event (generator.c:854-878, 917-946), and a line
 * without a sweep moves its position once per such block, all at once: advance_len
 * (sau/line.c:385-398) adds the block's length and, when that reaches `end`, restarts
 * the position at 0 *for the whole block* and drops LP_TIME; sauLine_run/sauLine_skip
 * (430-445, 456-473) do the same when a sweep's goal is reached inside a block -- the
 * rest of that block does not count. The position then keeps cycling block by block.
 * It is observed only by sauLine_copy at an event (305-309: `end -= pos`), where it
 * decides the length of a sweep that inherits its time. This backend renders segments
 * that are cut elsewhere (other programs' events, read-ahead runs spanning many host
 * calls, its own blocks), so a held line walks the reference's block lattice
 * explicitly: Lattice says where the reference's blocks lie within the segment, and
 * the line carries, in the upper bits of `flags`, the frames it has run in the
 * reference block that is still open (LPX_PEND) and whether that block is the one in
 * which its sweep ended (LPX_SKIP). */
constexpr uint32_t LAT_BLOCK = 1024;      /* generator.c:24 BUF_LEN */
constexpr uint32_t LPX_SKIP = 1u << 15;
constexpr uint32_t LPX_PEND_SHIFT = 16;   /* 11 bits: 0..1024 */
constexpr uint32_t LPX_PEND_MASK = 0x7ffu << LPX_PEND_SHIFT;

struct Lattice {
	uint32_t e0;        /* frames between the start of the reference's current span (call start or this
	                     * program's latest event, whichever is later) and the segment's first frame */
	uint32_t span_left; /* frames from the segment's first frame to that span's end (call end or this
	                     * program's next event) */
	uint32_t call_len;  /* length of the spans after it: the host's call size */
};
SAU_HD Lattice lattice_none() { Lattice l; l.e0 = 0; l.span_left = 0xffffffffu; l.call_len = 0xffffffffu; return l; }

/* Where the reference's calls end, for the loop tails of `cub` (sweep_cub_tail): the block being evaluated starts at frame
 * `off` of the segment; `rem` frames from there the operator, one of its ancestors or the voice stops (TAIL_FAR or more:
 * not within reach of any reference block that overlaps this one). */
constexpr uint32_t TAIL_FAR = 0xFFFFu;
struct TailCtx {
	Lattice lat;
	uint32_t ev_left; /* frames from the segment's start to the program's next event (~0u: none): blocks end there too */
	uint32_t off, rem, on;
};
SAU_HD TailCtx tail_none() { TailCtx c; c.lat = lattice_none(); c.ev_left = 0xffffffffu; c.off = 0; c.rem = TAIL_FAR; c.on = 0; return c; }
/* the reference block that holds sample j: frames already in it before the sample, frames from the sample to its end
 * (>= 1) -- the block of 1024 from the start of its span, ended by the span's end, the program's next event, and
 * where the operator (an ancestor, the voice) stops. 32-bit arithmetic throughout (frames stay far below 2^31). */
SAU_HD void ref_block_at(const TailCtx &c, uint32_t j, uint32_t &in_blk, uint32_t &blk_rem) {
	const uint32_t f = c.off + j;
	uint32_t sp, span_rem;
	if (f < c.lat.span_left) { sp = c.lat.e0 + f; span_rem = c.lat.span_left - f; }
	else {
		const uint32_t r = (f - c.lat.span_left) % c.lat.call_len;
		sp = r; span_rem = c.lat.call_len - r;
	}
	if (c.ev_left != 0xffffffffu && c.ev_left > f && c.ev_left - f < span_rem) span_rem = c.ev_left - f;
	in_blk = sp % LAT_BLOCK;
	blk_rem = LAT_BLOCK - in_blk;
	if (span_rem < blk_rem) blk_rem = span_rem;
	if (c.rem < TAIL_FAR && c.rem > j && c.rem - j < blk_rem) blk_rem = c.rem - j;
}
/* sample j of the block is the last of a sauLine_fill_cub call of odd length; goal_rem: frames of the sweep left at sample 0 */
SAU_HD bool cub_fill_is_tail(const TailCtx &c, uint32_t j, uint32_t goal_rem) {
	if (!c.on) return false;
	uint32_t in_blk, blk_rem;
	ref_block_at(c, j, in_blk, blk_rem);
	if (goal_rem > j && goal_rem - j < blk_rem) blk_rem = goal_rem - j;
	return blk_rem == 1 && (((in_blk + blk_rem) & 1u) != 0);
}
/* sample j of the block is among the last len % 4 of a sauLine_map_cub call */
SAU_HD bool cub_map_is_tail(const TailCtx &c, uint32_t j) {
	if (!c.on) return false;
	uint32_t in_blk, blk_rem;
	ref_block_at(c, j, in_blk, blk_rem);
	return in_blk >= ((in_blk + blk_rem) & ~3u);
}
/* (a real call for the time-parallel kernels, scalars only: inlined into their R branch once per row the predicate cost
 * them up to 130 spilled VGPRs) */
SAU_HD_CALL bool cub_map_is_tail_v(uint32_t e0, uint32_t span_left, uint32_t call_len, uint32_t ev_left, uint32_t rem, uint32_t j) {
	TailCtx c;
	c.lat.e0 = e0; c.lat.span_left = span_left; c.lat.call_len = call_len; c.ev_left = ev_left; c.off = 0; c.rem = rem; c.on = 1;
	return cub_map_is_tail(c, j);
}

/* advance_len (sau/line.c:385-398) for k >= 1 consecutive blocks of b >= 1 frames each, closed form */
SAU_HD void line_hold_blocks(LineState &o, uint32_t b, uint32_t k) {
	if (o.pos >= o.end) { /* restarts with the first block */
		o.pos = 0;
		o.flags &= ~LP_TIME;
		if (--k == 0 || o.end == 0) return;
	}
	/* blocks until the position reaches the end */
	const uint32_t left = o.end - o.pos;
	const uint32_t j = left / b + (left % b ? 1u : 0u);
	if (k < j) { o.pos += k * b; return; }
	o.flags &= ~LP_TIME;
	k -= j;
	const uint32_t q = o.end / b + (o.end % b ? 1u : 0u); /* period of the cycle from 0 */
	o.pos = (k % q) * b;
}

/* line_hold_blocks(o, LAT_BLOCK, k) and line_hold_blocks(o, b, 1) without their divisions (LAT_BLOCK is a power of two;
 * one block either fits before the end or wraps to 0) */
SAU_HD void line_hold_blocks_lat(LineState &o, uint32_t k) {
	static_assert((LAT_BLOCK & (LAT_BLOCK - 1)) == 0, "shifts below");
	constexpr uint32_t SH = 10;
	static_assert((1u << SH) == LAT_BLOCK, "LAT_BLOCK is 1024");
	if (o.pos >= o.end) {
		o.pos = 0;
		o.flags &= ~LP_TIME;
		if (--k == 0 || o.end == 0) return;
	}
	const uint32_t left = o.end - o.pos;
	const uint32_t j = (left + (LAT_BLOCK - 1)) >> SH;
	if (k < j) { o.pos += k << SH; return; }
	o.flags &= ~LP_TIME;
	k -= j;
	const uint32_t q = (o.end + (LAT_BLOCK - 1)) >> SH;
	o.pos = (k < q ? k : k % q) << SH;
}
SAU_HD void line_hold_block_one(LineState &o, uint32_t b) {
	if (o.pos >= o.end) {
		o.pos = 0;
		o.flags &= ~LP_TIME;
		return;
	}
	if (o.end - o.pos > b) { o.pos += b; return; } /* (k = 1 < j = ceil(left / b)) */
	o.flags &= ~LP_TIME;
	o.pos = 0; /* j = 1, k - j = 0: (0 % q) * b */
}

/* one reference block of `b` frames has ended for a held line */
SAU_HD void line_hold_block_end(LineState &o, uint32_t b) {
	if (o.flags & LPX_SKIP) o.flags &= ~LPX_SKIP; /* its sweep ended inside this block: sau/line.c:436 pos = 0 stands */
	else if (b) line_hold_blocks(o, b, 1);
}

/* what is still open belongs to a reference block that has ended since (the operator stopped
 * inside it, or an event cut it short): called at span starts and before sauLine_copy */
SAU_HD void line_lat_flush(LineState &o) {
	const uint32_t pend = (o.flags & LPX_PEND_MASK) >> LPX_PEND_SHIFT;
	if (!(o.flags & LP_GOAL) && (pend || (o.flags & LPX_SKIP))) line_hold_block_end(o, pend);
	o.flags &= ~(LPX_PEND_MASK | LPX_SKIP);
}

/* A held line runs n more frames, the first of them frame `off` of the segment. */
SAU_HD void line_hold_lat(LineState &o, uint32_t n, const Lattice &lat, uint32_t off) {
	uint32_t pend = (o.flags & LPX_PEND_MASK) >> LPX_PEND_SHIFT;
	o.flags &= ~LPX_PEND_MASK;
	while (true) {
		/* the span frame `off` lies in: position within it, frames to its end */
		uint32_t sp, rem;
		if (off < lat.span_left) { sp = lat.e0 + off; rem = lat.span_left - off; }
		else {
			const uint32_t r = (off - lat.span_left) % lat.call_len;
			sp = r; rem = lat.call_len - r;
		}
		if (sp == 0 && (pend || (o.flags & LPX_SKIP))) { line_hold_block_end(o, pend); pend = 0; }
		if (n == 0) break;
		/* whole spans in a row (a 60 s segment has 234 spans of the host's call size; walking them through the general
		 * code below, four integer divisions each, cost finalize_kernel 120 us per launch): every such span is the same
		 * run of reference blocks -- `full` of LAT_BLOCK frames, then one of `tail` -- and for those two shapes
		 * line_hold_blocks() needs no division but the one of a position that wraps */
		if (sp == 0 && off >= lat.span_left && n >= lat.call_len) {
			const uint32_t full = lat.call_len / LAT_BLOCK, tail = lat.call_len % LAT_BLOCK;
			do {
				/* whole spans that stay clear of the line's end: a span that begins at position p with p + call_len < end moves
				 * the position by call_len and nothing else (its full blocks and its last, shorter one all fit: line_hold_blocks_lat,
				 * line_hold_block_one), so k such spans move it by k call_len. A line's end is its operator's time, so a held line
				 * walks towards it for the whole of a long segment: BASELINE config 4's 60 s are 234 spans per line, finalize_kernel's
				 * long pole (59 us of a 4.2 ms step) until round 6 */
				if (o.pos < o.end && o.end - o.pos > lat.call_len) {
					uint32_t k = (o.end - o.pos - 1) / lat.call_len;
					if (k > n / lat.call_len) k = n / lat.call_len;
					if (k) {
						o.pos += k * lat.call_len; n -= k * lat.call_len; off += k * lat.call_len;
						continue;
					}
				}
				const uint32_t pos0 = o.pos, flags0 = o.flags;
				if (full) line_hold_blocks_lat(o, full);
				if (tail) line_hold_block_one(o, tail);
				n -= lat.call_len; off += lat.call_len;
				/* (a span that leaves the line where it was -- a line that was never swept sits at position 0 for good -- is one
				 * of a row of such spans: what a span does depends on the line's state alone. Round 6: a 60 s segment's 234 spans
				 * were finalize_kernel's long pole, 59 us per BASELINE config-4 step) */
				if (o.pos == pos0 && o.flags == flags0 && n >= lat.call_len) {
					const uint32_t k = n / lat.call_len;
					n -= k * lat.call_len; off += k * lat.call_len;
					break;
				}
			} while (n >= lat.call_len);
			if (n == 0) break;
			continue;
		}
		uint32_t m = n < rem ? n : rem; /* frames run within this span */
		const bool to_end = m == rem;
		n -= m; off += m;
		const uint32_t in_blk = sp % LAT_BLOCK;
		if (in_blk || pend || (o.flags & LPX_SKIP)) { /* a block that is already open */
			uint32_t b = LAT_BLOCK - in_blk;
			if (b > rem) b = rem;
			if (m < b) { pend += m; break; }
			line_hold_block_end(o, pend + b);
			pend = 0; m -= b;
		}
		const uint32_t full = m / LAT_BLOCK, tail = m % LAT_BLOCK;
		if (full) line_hold_blocks(o, LAT_BLOCK, full);
		if (tail) {
			if (to_end) line_hold_blocks(o, tail, 1); /* the span's last, shorter block */
			else { pend = tail; break; }
		}
		if (n == 0 && !to_end) break;
	}
	o.flags |= pend << LPX_PEND_SHIFT;
}

/* a sweep's goal was reached with frame `at` of the segment the next to come: unless a reference
 * block ends right there, the rest of the current one does not move the position */
SAU_HD void line_goal_end_lat(LineState &o, const Lattice &lat, uint32_t at) {
	uint32_t sp;
	bool span_end;
	if (at < lat.span_left) { sp = lat.e0 + at; span_end = false; }
	else { sp = (at - lat.span_left) % lat.call_len; span_end = sp == 0; }
	o.flags &= ~LPX_PEND_MASK;
	if (span_end || sp % LAT_BLOCK == 0) o.flags &= ~LPX_SKIP;
	else o.flags |= LPX_SKIP;
}

/* What one block of `len` samples of a line looks like: samples [0,goal_len)
 * follow the sweep, the rest hold a constant; each part optionally multiplied
 * by the ratio buffer. Produced by line_begin(), which also advances the state
 * exactly as sauLine_run does (sau/line.c:417-445 incl. sauLine_get 349-378). */
struct LineBlock {
	Sweep sw;
	uint32_t goal_len;
	bool mul_goal, mul_hold;
	float hold;
	uint32_t goal_rem; /* frames of the sweep left at the block's first sample, not cut at the block's length (TailCtx) */
};

/* The no-sweep part of sauLine_run: advance_len (sau/line.c:385-398). After it
 * the block holds v0 (times the ratio buffer when LP_STATE_RATIO). */
SAU_HD void line_advance_hold(LineState &o, uint32_t len, const Lattice &lat, uint32_t off) {
	line_hold_lat(o, len, lat, off);
}

/* have_mul: a ratio buffer exists; mul0: its first value (only read when the
 * state/goal ratio flags disagree, sau/line.c:358-370). `off`: the segment frame the
 * block starts at (see Lattice). */
SAU_HD LineBlock line_begin_body(LineState &o, uint32_t len, bool have_mul, float mul0,
		const Lattice &lat, uint32_t off) {
	LineBlock b;
	b.goal_len = 0; b.mul_goal = false; b.mul_hold = false; b.hold = 0.f; b.goal_rem = 0;
	b.sw = sweep_setup(LN_sah, 0.f, 0.f, 0, 1);
	bool hold;
	if (!(o.flags & LP_GOAL)) {
		line_hold_lat(o, len, lat, off);
		hold = true;
	} else {
		bool mul = have_mul;
		if (o.flags & LP_GOAL_RATIO) {
			if (!(o.flags & LP_STATE_RATIO)) {
				if (have_mul) o.v0 /= mul0;
				o.flags |= LP_STATE_RATIO;
			}
		} else {
			if (o.flags & LP_STATE_RATIO) {
				if (have_mul) o.v0 *= mul0;
				o.flags &= ~LP_STATE_RATIO;
			}
			mul = false;
		}
		uint32_t glen = 0;
		if (o.pos < o.end) {
			glen = o.end - o.pos;
			b.goal_rem = glen;
			if (glen > len) glen = len;
			b.sw = sweep_setup(o.type, o.v0, o.vt, o.pos, o.end);
			b.mul_goal = mul;
		}
		b.goal_len = glen;
		o.pos += glen;
		hold = (o.pos >= o.end);
		if (hold) {
			o.v0 = o.vt;
			o.pos = 0;
			o.flags &= ~(LP_GOAL | LP_GOAL_RATIO | LP_TIME);
			line_goal_end_lat(o, lat, off + glen);
			line_hold_lat(o, len - glen, lat, off + glen);
		}
	}
	if (hold) {
		b.mul_hold = have_mul && (o.flags & LP_STATE_RATIO);
		b.hold = o.v0;
	}
	return b;
}

SAU_HD_CALL LineBlock line_begin(LineState &o, uint32_t len, bool have_mul, float mul0,
		const Lattice &lat, uint32_t off) {
	return line_begin_body(o, len, have_mul, mul0, lat, off);
}

SAU_HD_CALL float line_value(const LineBlock &b, uint32_t i, float mul_i) {
	if (i < b.goal_len) {
		float v = sweep_value(b.sw, i);
		return b.mul_goal ? v * mul_i : v;
	}
	return b.mul_hold ? b.hold * mul_i : b.hold;
}
/* ... with the reference's loop tails (only `cub` has any) */
SAU_HD_CALL float line_value_t(const LineBlock &b, uint32_t i, float mul_i, const TailCtx &tc) {
	if (i < b.goal_len && b.sw.type == LN_cub && cub_fill_is_tail(tc, i, b.goal_rem)) {
		const float v = sweep_cub_tail(b.sw, i);
		return b.mul_goal ? v * mul_i : v;
	}
	return line_value(b, i, mul_i);
}

/* The same two for callers that keep line states and blocks in registers: arguments and
 * results by value (a reference parameter of a real call pins the caller's object in scratch
 * memory, and every later use of it becomes a memory round trip -- measured at about 2 us per
 * plan step in the block loop). line_begin() = line_block_v() for the block +
 * line_begin_state() for the state it leaves behind. */
SAU_HD_CALL LineBlock line_block_v(LineState o, uint32_t len, bool have_mul, float mul0) {
	return line_begin_body(o, len, have_mul, mul0, lattice_none(), 0);
}
SAU_HD void line_begin_state(LineState &o, uint32_t len, bool have_mul, float mul0,
		const Lattice &lat, uint32_t off) {
	if (!(o.flags & LP_GOAL)) {
		line_hold_lat(o, len, lat, off);
		return;
	}
	if (o.flags & LP_GOAL_RATIO) {
		if (!(o.flags & LP_STATE_RATIO)) {
			if (have_mul) o.v0 /= mul0;
			o.flags |= LP_STATE_RATIO;
		}
	} else if (o.flags & LP_STATE_RATIO) {
		if (have_mul) o.v0 *= mul0;
		o.flags &= ~LP_STATE_RATIO;
	}
	uint32_t glen = 0;
	if (o.pos < o.end) {
		glen = o.end - o.pos;
		if (glen > len) glen = len;
	}
	o.pos += glen;
	if (o.pos >= o.end) {
		o.v0 = o.vt;
		o.pos = 0;
		o.flags &= ~(LP_GOAL | LP_GOAL_RATIO | LP_TIME);
		line_goal_end_lat(o, lat, off + glen);
		line_hold_lat(o, len - glen, lat, off + glen);
	}
}
SAU_HD_CALL float line_value_v(LineBlock b, uint32_t i, float mul_i) {
	if (i < b.goal_len) {
		float v = sweep_value_inl(b.sw, i);
		return b.mul_goal ? v * mul_i : v;
	}
	return b.mul_hold ? b.hold * mul_i : b.hold;
}
/* ... with the reference's loop tails (only `cub` has any); the block loop's form of line_value_t */
SAU_HD_CALL float line_value_vt(LineBlock b, uint32_t i, float mul_i, TailCtx tc) {
	if (tc.on && i < b.goal_len && b.sw.type == LN_cub && cub_fill_is_tail(tc, i, b.goal_rem)) {
		const float v = sweep_cub_tail(b.sw, i);
		return b.mul_goal ? v * mul_i : v;
	}
	return line_value_v(b, i, mul_i);
}

/* sau/line.c:456-473 */
SAU_HD void line_skip(LineState &o, uint32_t len, const Lattice &lat, uint32_t off) {
	if (!(o.flags & LP_GOAL)) {
		line_hold_lat(o, len, lat, off);
		return;
	}
	uint32_t l = 0;
	if (o.pos < o.end) {
		l = o.end - o.pos;
		if (l > len) l = len;
		o.pos += l;
	}
	if (o.pos >= o.end) {
		o.pos = 0;
		o.flags &= ~LP_TIME;
		o.v0 = o.vt;
		if (o.flags & LP_GOAL_RATIO) o.flags |= LP_STATE_RATIO;
		else o.flags &= ~LP_STATE_RATIO;
		o.flags &= ~(LP_GOAL | LP_GOAL_RATIO);
		line_goal_end_lat(o, lat, off + l);
		line_hold_lat(o, len - l, lat, off + l);
	}
}

/* An update as carried by an event (sauLine by value). sau/line.c:287-332.
 * end_samples = time_ms converted with the generator's sample rate. */
struct LineUpdate {
	float v0, vt;
	uint32_t end_samples;
	uint32_t type;
	uint32_t flags; /* LP_*; 0 = no update */
};

SAU_HD void line_copy(LineState &o, const LineUpdate &src, bool loop_tails = false) {
	if (!src.flags)
		return;
	line_lat_flush(o); /* an event ends the reference's block */
	uint32_t mask = 0;
	if (src.flags & LP_STATE) {
		o.v0 = src.v0;
		mask |= LP_STATE | LP_STATE_RATIO;
	} else if (o.flags & LP_GOAL) {
		if (src.flags & LP_GOAL) {
			/* sauLine_get(o, &f, 1, NULL): value at the current position */
			LineState t = o;
			LineBlock b = line_begin(t, 1, false, 0.f, lattice_none(), 0);
			/* line_begin advanced a copy; only the ratio-flag side effect
			 * of sauLine_get persists in the reference (v0 is untouched
			 * without a ratio buffer). */
			if (o.flags & LP_GOAL_RATIO) o.flags |= LP_STATE_RATIO;
			else o.flags &= ~LP_STATE_RATIO;
			if (b.goal_len > 0) /* (a fill of length 1: for `cub` the reference's scalar loop tail, sweep_cub_tail) */
				o.v0 = (loop_tails && b.sw.type == LN_cub) ? sweep_cub_tail(b.sw, 0) : sweep_value(b.sw, 0);
			/* else: sauLine_get wrote nothing; reference then reads an
			 * uninitialised float -- unreachable in practice because a set
			 * goal always has pos < end between blocks. */
		}
	}
	if (src.flags & LP_GOAL) {
		o.vt = src.vt;
		if (src.flags & LP_TIME_IF_NEW)
			o.end -= o.pos;
		o.pos = 0;
		mask |= LP_GOAL | LP_GOAL_RATIO;
	}
	if (src.flags & LP_TYPE) {
		o.type = src.type;
		mask |= LP_TYPE;
	}
	if (!(o.flags & LP_TIME) || !(src.flags & LP_TIME_IF_NEW)) {
		if (src.flags & LP_TIME) {
			o.end = src.end_samples;
			mask |= LP_TIME;
		}
	}
	o.flags &= ~mask;
	o.flags |= (src.flags & mask);
}

/* ---- wave oscillator ------------------------------------------------------ */

/* Hermite coefficients of one table index, precomputed from the four taps in
 * the exact expression order of sauWave_get_herp (sau/wave.h:127-141). c0 and
 * c1 are exactly representable as f32; c2 and c3 need f64.
 *
 * The reference evaluates ((c3*x + c2)*x + c1)*x + c0 with x = frac * 2^-21
 * (frac: the low 21 phase bits).  Stored here are c3 * 2^-63, c2 * 2^-42 and
 * c1 * 2^-21, and the polynomial runs on X = (double)frac: every intermediate
 * is the reference's times a power of two, so each rounding is the same and
 * the final sum (scale 2^0) is bit-identical, with two conversions fewer. */
struct HerpC23 { double c3, c2; };
struct HerpC01 { float c1, c0; };

SAU_HD void herp_coeffs(const float *lut, uint32_t ind, HerpC23 &hi, HerpC01 &lo) {
	float s0 = lut[(ind - 1) & WAVE_MASK];
	float s1 = lut[ind & WAVE_MASK];
	float s2 = lut[(ind + 1) & WAVE_MASK];
	float s3 = lut[(ind + 2) & WAVE_MASK];
	lo.c0 = s1;
	lo.c1 = (float)(1 / 2.0 * (double)(s2 - s0)) * 0x1p-21f; /* halving and scaling are exact */
	hi.c2 = ((double)s0 - 5 / 2.0 * (double)s1 + (double)(2 * s2) - 1 / 2.0 * (double)s3) * 0x1p-42;
	hi.c3 = (1 / 2.0 * (double)(s3 - s0) + 3 / 2.0 * (double)(s1 - s2)) * 0x1p-63;
}
/* false if scaling c1 would leave the normal f32 range (inexact): such a
 * table cannot use the prescaled form */
SAU_HD bool herp_c1_scalable(const float *lut, uint32_t ind) {
	float d = lut[(ind + 1) & WAVE_MASK] - lut[(ind - 1) & WAVE_MASK];
	float a = d < 0 ? -d : d;
	return a == 0.f || a >= 0x1p-100f;
}

SAU_HD double herp_poly(const HerpC23 &hi, const HerpC01 &lo, uint32_t phase) {
	double x = (double)(phase & (SLEN - 1));
	return ((hi.c3 * x + hi.c2) * x + (double)lo.c1) * x + (double)lo.c0;
}

/* herp_poly() without its table value: herp_poly(hi, lo, p) == herp_poly_rise(hi, lo, p) + (double)lo.c0, the last step */
SAU_HD double herp_poly_rise(const HerpC23 &hi, const HerpC01 &lo, uint32_t phase) {
	double x = (double)(phase & (SLEN - 1));
	return ((hi.c3 * x + hi.c2) * x + (double)lo.c1) * x;
}
/* wosc.h:215-231, the sample a (re)started oscillator begins with, as the reference build computes it: sauWOsc_reset is a
 * function of its own there, and gcc's fast-math takes the Hermite value one table step back apart -- (Is - y1') - P', its
 * table value y1' and its polynomial part P' subtracted one after the other, not their rounded sum (objdump of
 * oracle/_ref/generator.o; an ulp now and then, which reaches the PCM through feedback plus a running sum: DESIGN.md 9, 8b).
 * Is0: herp_poly at the first phase; rise_p, c0_p: herp_poly_rise and lo.c0 at that phase - SLEN. */
SAU_HD float wosc_reset_s(double Is0, double rise_p, float c0_p, float diff_scale, float diff_offset);

/* Correctly rounded a/b for operands well inside the normal range (here
 * a ~ 1e8..1e9 and 1 <= |b| <= 2^31): the reciprocal-refinement sequence the
 * compiler itself emits for IEEE division, without the range scaling and the
 * special-case fix-up that these operands never need. */
SAU_HD float div_f32_normal(float a, float b) {
#if defined(__HIP_DEVICE_COMPILE__)
	float r = __builtin_amdgcn_rcpf(b);
	float e = fmaf(-b, r, 1.0f);
	r = fmaf(e, r, r);
	float q = a * r;
	float rem = fmaf(-b, q, a);
	q = fmaf(rem, r, q);
	rem = fmaf(-b, q, a);
	return fmaf(rem, r, q);
#else
	return a / b;
#endif
}

/* The differentiator's division, diff_scale / (float)dphase (wosc.h:253): the
 * dividend is one of twelve per-wave constants and the divisor an integer of
 * magnitude 1..2^31 rounded to f32.  For exactly those operands gfx950's
 * v_rcp_f32 followed by ONE residual correction is already correctly rounded
 * -- not in general: tests/test_gpu_units.py::test_differentiator_division_exhaustive
 * compares it with IEEE division for every such divisor (2.1e9 of them) and
 * every wave's constant on the device (0 mismatches; the uncorrected product
 * a * rcp(b) differs in 7 % of the cases). */
SAU_HD float div_diff_scale(float a, float b) {
#if defined(__HIP_DEVICE_COMPILE__)
	float r = __builtin_amdgcn_rcpf(b);
	float q = a * r;
	return fmaf(fmaf(-b, q, a), r, q);
#else
	return a / b;
#endif
}

/* wosc.h:250-256: one differentiated output sample */
SAU_HD float wosc_diff(double Is, double prev_Is, int32_t phase_diff,
		float diff_scale, float diff_offset) {
	double x = (double)div_diff_scale(diff_scale, (float)phase_diff);
	return (float)((Is - prev_Is) * x + (double)diff_offset);
}

SAU_HD float wosc_reset_s(double Is0, double rise_p, float c0_p, float diff_scale, float diff_offset) {
	const double x = (double)div_diff_scale(diff_scale, (float)(int32_t)SLEN);
	return (float)(((Is0 - (double)c0_p) - rise_p) * x + (double)diff_offset);
}

/* Phase offset from the PM inputs, wosc.h:135-169 / rasg.h:165-222 in
 * ref-build forms. phase_scale is 2^31 (W, R half-shape) or 2^32 (R rate2x). */
constexpr float FPM_SCALE = (float)(1.0 / 632.45553203367586639978);
SAU_HD int64_t pm_offset(bool has_pm, bool has_fpm, float pm, float fpm, float f,
		float phase_scale) {
	if (has_pm && has_fpm) {
		float p = pm + ((fpm * f) * FPM_SCALE);
		return rint64(p * phase_scale);
	} else if (has_pm) {
		return rint64(pm * phase_scale);
	} else if (has_fpm) {
		return rint64((fpm * f) * (FPM_SCALE * phase_scale));
	}
	return 0;
}

/* The same for the W oscillator, where only the low 32 bits matter
 * (phase_scale = 2^31). */
SAU_HD uint32_t pm_offset32(bool has_pm, bool has_fpm, float pm, float fpm, float f) {
	if (has_pm && has_fpm) {
		float p = pm + ((fpm * f) * FPM_SCALE);
		return rint32w_p31(p);
	} else if (has_pm) {
		return rint32w_p31(pm);
	} else if (has_fpm) {
		return rint32w((fpm * f) * (FPM_SCALE * 0x1p31f));
	}
	return 0;
}

/* ---- noise ---------------------------------------------------------------- */

/* noise.h:61-70 */
SAU_HD float soft_sqrtm2logp1(float x) {
	const float k0 = -0.80270565422983103084f;
	const float k1 = +5.52274428214641442648f;
	const float k2 = -138.87126103150588693697f;
	float x2 = x * x;
	float x4 = x2 * x2;
	return 0.5f + x * (k0 + x4 * (k1 + x4 * k2));
}
/* noise.h:77-81 */
SAU_HD float ssgauss_dist4(float x) {
	float x2 = x * x;
	float gx = (x + x2) * 0.5f;
	return x * (1 - gx * (1 - x2));
}
/* noise.h:90-98 */
SAU_HD float franssgauss32(uint32_t n) {
	int32_t s0 = (int32_t)ranfast32(n);
	int32_t s1 = (int32_t)mcg32((uint32_t)s0);
	/* (the reference's `float a = s0 * 0x1p-32;` multiplies in double and rounds once: the nearest float to s0 x 2^-32. So is
	 * (float)s0 x 2^-32 -- the conversion rounds the same 32-bit integer to the same 24 bits, and the power of two moves only the
	 * exponent, far from the denormals: identical bits, without the two f64 conversions and the f64 scaling, 4 cycles of issue each) */
	float a = (float)s0 * 0x1p-32f;
	float b = (float)s1 * 0x1p-32f;
	float c = ssgauss_dist4(soft_sqrtm2logp1(a));
	b = c * sinpi_d5f(b);
	return b;
}

/* Sample with counter value n of the counter-hash noises (noise.h:41-128).
 * `re` `vi` `bv` carry state and are handled by the callers:
 *   re: running u32 sum of (int32)hash >> 6, folded       (noise.h:136-147)
 *   vi: hash(n)/2 - hash(n-1)/2                             (noise.h:149-159)
 *   bv: tw-like value minus its predecessor                  (noise.h:161-172) */
SAU_HD float noise_stateless(uint32_t type, uint32_t n) {
	switch (type) {
	default:
	case NZ_wh: return fscalei(ranfast32(n), 0x1p-31f);
	case NZ_gw: return franssgauss32(n);
	case NZ_bw: return (float)(sar32((int32_t)ranfast32(n), 31) * 2 + 1);
	case NZ_tw: {
		int32_t s = sar32((int32_t)ranfast32(n), 31) * 2 + 1;
		return (n & 1u) ? (float)s : 0.f;
	}
	}
}
SAU_HD int32_t noise_bv_term(uint32_t n) {
	int32_t s1 = sar32((int32_t)ranfast32(n), 31);
	return (n & 1u) ? (s1 * 2 + 1) : 0;
}

/* ---- random-segments oscillator -------------------------------------------- */

struct RasParams {
	uint32_t func, flags, level, alpha, line;
	float vbin_scale; /* rasg.h:398-402 */
	float perlin_amp; /* rasg.h:244-247,702-705 */
};

SAU_HD uint32_t ras_level9() { return (9 - 4) * (9 - 4) + 2; } /* program.h:146-148 */

SAU_HD float perlin_amp_of(uint32_t line) { /* sau/line.h:18-32 */
	switch (line) {
	case LN_sah: case LN_uwh: return 1.f;
	case LN_exp: case LN_log: case LN_xpe: case LN_lge: return 1.55845810035f;
	case LN_sqe: case LN_nhl: return 1.89339094650f;
	default: return 2.f;
	}
}

SAU_HD RasParams ras_params(uint32_t func, uint32_t flags, uint32_t level,
		uint32_t alpha, uint32_t line) {
	RasParams p;
	p.func = func; p.flags = flags; p.level = level; p.alpha = alpha; p.line = line;
	const float scale_diff = 1.f - ((float)sar32(INT32_MAX, (int)level) / 0x1p31f);
	p.vbin_scale = (1.f + scale_diff * scale_diff) / 0x1p31f;
	p.perlin_amp = (flags & (RO_HALFSHAPE | RO_ZIGZAG)) ? 1.f : perlin_amp_of(line);
	return p;
}

/* Segment end values for a cycle index; rasg.h:299-671. */
SAU_HD void ras_ends(const RasParams &c, uint32_t cycle, float &a, float &b) {
	const int sr = (int)c.level;
	const bool violet = (c.flags & RO_VIOLET) != 0;
	switch (c.func) {
	default:
	case RF_URAND:
		if (violet) {
			uint32_t s0 = ranfast32(cycle - 1) / 2;
			uint32_t s1 = ranfast32(cycle) / 2;
			uint32_t s2 = ranfast32(cycle + 1) / 2;
			a = fscalei(s1 - s0, 0x1p-31f);
			b = fscalei(s2 - s1, 0x1p-31f);
		} else {
			a = fscalei(ranfast32(cycle), 0x1p-31f);
			b = fscalei(ranfast32(cycle + 1), 0x1p-31f);
		}
		break;
	case RF_GAUSS:
		a = franssgauss32(cycle);
		b = franssgauss32(cycle + 1);
		break;
	case RF_BIN:
		if (violet) {
			uint32_t sb = (cycle & 1u) << 31;
			uint32_t sb_flip = (1u << 31) - sb;
			uint32_t s0 = (uint32_t)divi((uint32_t)sar32((int32_t)ranfast32(cycle - 1), sr) + sb, 2);
			uint32_t s1 = (uint32_t)divi((uint32_t)sar32((int32_t)ranfast32(cycle), sr) + sb_flip, 2);
			uint32_t s2 = (uint32_t)divi((uint32_t)sar32((int32_t)ranfast32(cycle + 1), sr) + sb, 2);
			a = fscalei(s1 - s0, c.vbin_scale);
			b = fscalei(s2 - s1, c.vbin_scale);
		} else {
			uint32_t offs = (uint32_t)INT32_MAX + (cycle & 1u) * 2u;
			uint32_t s1 = (uint32_t)sar32((int32_t)ranfast32(cycle), sr) + offs;
			uint32_t s2 = (uint32_t)sar32((int32_t)ranfast32(cycle + 1), sr) - offs;
			a = fscalei(s1, 0x1p-31f);
			b = fscalei(s2, 0x1p-31f);
		}
		break;
	case RF_TERN: {
		uint32_t sb = (cycle & 1u) << 31;
		uint32_t sb_flip = (1u << 31) - sb;
		uint32_t s1 = (uint32_t)sar32((int32_t)ranfast32(cycle), sr) + sb_flip;
		uint32_t s2 = (uint32_t)sar32((int32_t)ranfast32(cycle + 1), sr) + sb;
		a = fscalei(s1, 0x1p-31f);
		b = fscalei(s2, 0x1p-31f);
		break;
	}
	case RF_FIXED:
		if (c.level >= ras_level9()) {
			a = (float)odd_sign(cycle);
			b = -a;
		} else if (violet) {
			uint32_t sign = (uint32_t)odd_sign(cycle);
			uint32_t s0 = (uint32_t)divi(sign * ((ranfast32(cycle - 1) >> sr) - (uint32_t)INT32_MAX), 2);
			uint32_t s1 = (uint32_t)divi((0u - sign) * ((ranfast32(cycle) >> sr) - (uint32_t)INT32_MAX), 2);
			uint32_t s2 = (uint32_t)divi(sign * ((ranfast32(cycle + 1) >> sr) - (uint32_t)INT32_MAX), 2);
			a = fscalei(s1 - s0, 0x1p-31f);
			b = fscalei(s2 - s1, 0x1p-31f);
		} else {
			uint32_t sign = (uint32_t)odd_sign(cycle);
			a = fscalei((0u - sign) * ((ranfast32(cycle) >> sr) - (uint32_t)INT32_MAX), 0x1p-31f);
			b = fscalei(sign * ((ranfast32(cycle + 1) >> sr) - (uint32_t)INT32_MAX), 0x1p-31f);
		}
		break;
	case RF_ADDREC: {
		uint32_t s0 = cycle * c.alpha;
		uint32_t s1 = (cycle + 1) * c.alpha;
		a = fscalei(s0, 0x1p-31f);
		b = fscalei(s1, 0x1p-31f);
		break;
	}
	}
}

/* One output value from cycle index and in-cycle phase in [0,1): rasg.h:692-743 (block form,
 * `block` true) or rasg.h:254-275 (per-sample form of the feedback loop). The two differ in the
 * reference build only in how -ffast-math associated the Perlin scaling: the block loop (vector
 * body and scalar tail alike) computes (a * phase) * amp and (b * amp) * (phase - 1), the six
 * per-sample loops keep the source's a * (amp * phase) and b * (amp * (phase - 1)). */
template <bool INL = false>
SAU_HD float ras_sample(const RasParams &c, uint32_t cycle, float phase, bool block, bool cub_tail = false) {
	float a, b;
	ras_ends(c, cycle, a, b);
	if (c.flags & RO_PERLIN) {
		if (block) {
			a = (a * phase) * c.perlin_amp;
			b = (b * c.perlin_amp) * (phase - 1.f);
		} else {
			a *= c.perlin_amp * phase;
			b *= c.perlin_amp * (phase - 1.f);
		}
	}
	if (c.flags & RO_HALFSHAPE) {
		/* sau_maxf / sau_minf (sau/math.h:121-130) as the build has them. Block loop (maxps b,a / minps a,b in the
		 * body, min/max b,a in the tails -- they part only for NaN and signed-zero pairs, which the block form's
		 * finite ends and phases in [0, 1) cannot produce); feedback loops: maxss a,b / minss b,a -- an unordered
		 * pair comes out swapped */
		float mx = block ? (a < b ? b : a) : (a > b ? a : b);
		float mn = block ? (a > b ? b : a) : (b < a ? b : a);
		a = mx; b = mn;
	}
	if (c.flags & RO_ZIGZAG) {
		float t = a; a = b; b = t;
	}
	if (c.flags & RO_SQUARE) {
		a *= fabsf(a);
		b *= fabsf(b);
	}
	if (cub_tail && c.line == LN_cub) return shape_cub_tail(phase, a, b); /* (the last 1-3 samples of the reference's block: TailCtx) */
	return INL ? shape_val_inl(c.line, phase, a, b) : shape_val(c.line, phase, a, b);
}

/* Split the 64-bit cycle|phase counter: rasg.h:184-186 */
SAU_HD void ras_split(uint64_t cp, uint32_t &cycle, float &phase) {
	cycle = (uint32_t)(cp >> 32);
	uint32_t ph = ((uint32_t)cp) >> 1;
	phase = (float)(int32_t)ph * 0x1p-31f;
}

/* The per-sample form of ras_sample() (block false, no loop tail) from segment ends the caller already has: the feedback loop's
 * chain wave keeps a lane's ends while its cycle stands (rchain_kernel, round 6) -- they are a function of the cycle alone
 * (rasg.h:299-671), and an oscillator of a few hundred Hz stays in one cycle for a hundred samples and more. */
SAU_HD float ras_sample_ends(const RasParams &c, float a, float b, float phase) {
	if (c.flags & RO_PERLIN) {
		a *= c.perlin_amp * phase;
		b *= c.perlin_amp * (phase - 1.f);
	}
	if (c.flags & RO_HALFSHAPE) { /* (maxss a,b / minss b,a: see ras_sample) */
		float mx = (a > b ? a : b);
		float mn = (b < a ? b : a);
		a = mx; b = mn;
	}
	if (c.flags & RO_ZIGZAG) {
		float t = a; a = b; b = t;
	}
	if (c.flags & RO_SQUARE) {
		a *= fabsf(a);
		b *= fabsf(b);
	}
	return shape_val_inl(c.line, phase, a, b);
}

/* ---- mixing ----------------------------------------------------------------- */

/* generator.c:384-426 */
SAU_HD float mix_combine(float dst, float in, float amp, bool wave_env, bool layer) {
	if (wave_env) {
		float s_amp = amp * 0.5f;
		float s = (in * s_amp) + fabsf(s_amp);
		return layer ? dst * s : s;
	}
	return layer ? dst + in * amp : in * amp;
}

SAU_HD float clampf(float x, float lo, float hi) { /* sau/math.h:133-137 */
	x = x < lo ? lo : x;
	x = x > hi ? hi : x;
	return x;
}

/* generator.c:803,822-823 */
/* player/sndfile.c:160-168: one sample in the other byte order */
SAU_HD int16_t pcm_swap(int16_t v) {
	uint16_t s = (uint16_t)v;
	return (int16_t)(uint16_t)((s >> 8) | (s << 8));
}
SAU_HD int16_t pcm16(float s) {
	/* a NaN in the mix (feedback that ran to infinity, inf - inf: absurd amounts only) leaves the reference build's clamp as
	 * its lower bound -- gcc's fast-math form of sau_fclampf is minss(maxss(s, -1), 1), and maxss hands back its second
	 * operand when the first is a NaN -- and so as -32767, in the loops' bodies and tails alike */
	if (s != s) s = -1.f;
	s = clampf(s, -1.f, 1.f);
#if defined(__HIP_DEVICE_COMPILE__)
	return (int16_t)__float2int_rn(s * (float)INT16_MAX);
#else
	return (int16_t)lrintf(s * (float)INT16_MAX);
#endif
}

} /* namespace saudev */
#endif /* SAU_DEV_MATH_H */
