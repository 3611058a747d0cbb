/* sau_oracle.c -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.
 *
 * A scalar CPU restatement of the saugns audio generator hot path, used only
 * as the checker for the MI355X backend (tests/, __graft_entry__.smoke() and
 * bench.py's cpu_baseline leg).  Nothing in saugns_amd/ may call into it.
 *
 * It follows the reference algorithm function by function (citations are
 * file:line in saugns v0.4.7) but is written for this repository: one flat
 * operator record instead of the reference's node union, explicit block
 * context, own table builder.  Strict C evaluation order, no FMA contraction,
 * no fast-math (compile with -ffp-contract=off).  The ramp shapes that the
 * reference's `-O3 -ffast-math` build of sau/line.c evaluates in a
 * re-associated order are restated in that order (see ramp_fill_*), so that
 * this oracle is bit-identical to oracle/_ref/libsau_ref.so -- that is what
 * "pinned" means here: tests/test_oracle_vs_ref.py checks it against the
 * compiled reference and tests/golden/ holds the reference's own outputs.
 */
#include "sau_abi.h"
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#define ORA_API __attribute__((visibility("default")))

/* ---------------------------------------------------------------------- */
/* scalar helpers (sau/math.h)                                             */
/* ---------------------------------------------------------------------- */

/* sau/math.h:35-46 */
static uint64_t ms_to_samples(uint64_t ms, uint64_t srate, int *carry) {
	uint64_t t = ms * srate;
	if (carry) {
		t += *carry;
		*carry = (int)(t % 1000);
	}
	return t / 1000;
}

/* sau/math.h:63-64 with generator.c:17: round-half-even, wraps via uint32 cast */
static inline int64_t rint_i64(float x) { return llrintf(x); }

/* sau/math.h:297-303 */
static inline uint32_t ranfast32(uint32_t n) {
	uint32_t s = n * 0x9e3779b9u;
	s ^= s >> 14;
	s = (s | 1) * s;
	s ^= s >> 13;
	return s;
}
/* sau/math.h:283-285 */
static inline uint32_t mcg32(uint32_t seed) { return seed * 0xe47135u; }
/* sau/math.h:94-96 */
static inline int32_t sar32(int32_t x, int s) {
	return x < 0 ? ~(~x >> s) : x >> s;
}
/* sau/math.h:112-118 */
static inline int32_t foldhd32(int32_t x) {
	uint32_t s = (uint32_t)x;
	if (s + (1u << 29) > (1u << 31))
		s = (1u << 31) + (1u << 30) - s;
	s = (s - (1u << 29)) * 2;
	return (int32_t)s;
}
/* sau/math.h:89-91 */
static inline int odd_sign(int n) { return 1 - ((n & 1) * 2); }
/* sau/math.h:366-379 */
static inline float sinpi_d5f(float x) {
	const float k0 = +3.14042741234069229463;
	const float k1 = -5.13655757476162831091;
	const float k2 = +2.29939170159543653372;
	float x2 = x * x;
	return x * (k0 + x2 * (k1 + x2 * k2));
}
/* generator.c:19 */
static inline float fscalei(uint32_t i, float scale) {
	return ((int32_t)i) * scale;
}
/* generator.c:20 */
static inline int32_t divi(uint32_t i, int32_t d) { return ((int32_t)i) / d; }

/* ---------------------------------------------------------------------- */
/* wave tables (sau/wave.c, sau/wave.h)                                    */
/* ---------------------------------------------------------------------- */

enum { WLEN = 2048, WMASK = WLEN - 1, WHALF = WLEN / 2, WQUART = WLEN / 4 };
enum { SLEN_BITS = 21 };

typedef struct WaveCoeffs { float amp_scale, amp_dc; int32_t phase_adj; } WaveCoeffs;

/* sau/wave.h:33-69 (numeric constants of the PILUT set) */
static const WaveCoeffs g_picoeffs[SAU_WAVE_NAMED] = {
	/* sin */ {1.27324153848, 0.0, (INT32_MIN / 2)},
	/* tri */ {1.00097751711, 0.0, 0},
	/* srs */ {1.52547437578, 0.0, 0},
	/* sqr */ {2.00000000000, 0.0, (INT32_MIN / 2)},
	/* ean */ {1.20275515347, -0.24257955076, 0},
	/* cat */ {1.37070880305, -0.23725526633, 0},
	/* eto */ {1.26113986272 * -1, 0.0, -(INT32_MIN / 2)},
	/* par */ {1.02639326795, -0.33333333333, 0},
	/* mto */ {1.57268451738, -0.23724704918, 0},
	/* saw */ {1.00048851979 * -1, 0.0, -(INT32_MIN / 2)},
	/* hsi */ {1.40333871035, -0.36334126990, 0},
	/* spa */ {1.07213756312, 0.27322393756, 0},
};

static float g_pilut[SAU_WAVE_NAMED][WLEN];
static int g_tables_ready;

/* sau/wave.c:77-98 -- integrate a table, centre it and scale to +/- scale */
static void integrate_table(float *dst, const float *src, float scale) {
	const float ivscale = 1.f / (WLEN * 0.125f);
	double dc = 0.f;
	for (int i = 0; i < WLEN; ++i) dc += src[i];
	dc /= WLEN;
	double acc = 0.f;
	float lo = 0.f, hi = 0.f;
	for (int i = 0; i < WLEN; ++i) {
		acc += src[i] - dc;
		float x = acc * ivscale;
		if (x < lo) lo = x;
		if (x > hi) hi = x;
		dst[i] = x;
	}
	float out_scale = scale / ((hi - lo) * 0.5f);
	float out_dc = -(hi + lo) * 0.5f;
	for (int i = 0; i < WLEN; ++i)
		dst[i] = (dst[i] + out_dc) * out_scale;
}

/* sau/wave.c:105-221, strict-order restatement. Four of the twelve tables
 * (srs ean cat mto, and eto's which is ean) differ by 1 ulp from what the
 * reference's fast-math build of wave.c produces; tests load
 * the reference's tables through ora_set_piluts() when exactness is wanted. */
static void build_tables(void) {
	static float sin_t[WLEN], tri_t[WLEN], pitri[WLEN], ean_t[WLEN], par_t[WLEN];
	static float srs_t[WLEN], cat_t[WLEN], mto_t[WLEN], hsi_t[WLEN], spa_t[WLEN];
	const float vs = 1.f;
	for (int i = 0; i < WHALF; ++i) {
		const double x = i * (1.f / WHALF);
		const float sin_x = sin(3.14159265358979323846 * x);
		sin_t[i] = vs * sin_x;
		sin_t[i + WHALF] = -vs * sin_x;
		const float srs_x = sqrtf(sin_x);
		srs_t[i] = vs * srs_x;
		hsi_t[i] = vs * (sin_x * 2 - 1.f);
		mto_t[i] = vs * (srs_x * 2 - 1.f);
		const float spa_x = sin(3.14159265358979323846 * 0.5f * (1 + x));
		spa_t[i + WQUART] = vs * (spa_x * 2 - 1.f);
	}
	for (int i = 0; i < WHALF; ++i) {
		const double x_rev = (WHALF - i) * (1.f / WHALF);
		par_t[i + WQUART] = vs * ((x_rev * x_rev) * 2.f - 1.f);
	}
	par_t[WHALF + WQUART] = -vs;
	spa_t[WHALF + WQUART] = -vs;
	for (int i = 0; i < WQUART; ++i) {
		const double x = i * (1.f / WQUART);
		const double x_rev = (WQUART - i) * (1.f / WQUART);
		pitri[i] = vs * ((x * x) - 1.f);
		pitri[i + WQUART] = vs * (1.f - (x_rev * x_rev));
		tri_t[i] = vs * x;
		tri_t[i + WQUART] = vs * x_rev;
		par_t[i] = par_t[WHALF - i];
		par_t[i + WHALF + WQUART] = par_t[WHALF + WQUART - i];
		spa_t[i] = spa_t[WHALF - i];
		spa_t[i + WHALF + WQUART] = spa_t[WHALF + WQUART - i];
	}
	for (int i = WHALF; i < WLEN; ++i) {
		pitri[i] = -pitri[i - WHALF];
		tri_t[i] = -tri_t[i - WHALF];
		hsi_t[i] = -vs;
		mto_t[i] = -vs;
		srs_t[i] = -srs_t[i - WHALF];
	}
	const float ean_dc_adj = (1.14603185654 - 1.f) / 2.f;
	const float ean_scale_adj = vs / 1.07301592827;
	for (int i = 0; i < WLEN; ++i) {
		ean_t[i] = (sin_t[i] + par_t[i] - tri_t[i] + ean_dc_adj) * ean_scale_adj;
		cat_t[i] = sin_t[i] + mto_t[i] - srs_t[i];
	}
	/* PILUT assignment, sau/wave.c:49-62 */
	memcpy(g_pilut[SAU_WAVE_N_sin], sin_t, sizeof sin_t);
	memcpy(g_pilut[SAU_WAVE_N_tri], pitri, sizeof pitri);
	integrate_table(g_pilut[SAU_WAVE_N_srs], srs_t, vs);
	memcpy(g_pilut[SAU_WAVE_N_sqr], tri_t, sizeof tri_t);
	integrate_table(g_pilut[SAU_WAVE_N_ean], ean_t, vs);
	integrate_table(g_pilut[SAU_WAVE_N_cat], cat_t, vs);
	memcpy(g_pilut[SAU_WAVE_N_eto], ean_t, sizeof ean_t);
	integrate_table(g_pilut[SAU_WAVE_N_par], par_t, vs);
	integrate_table(g_pilut[SAU_WAVE_N_mto], mto_t, vs);
	memcpy(g_pilut[SAU_WAVE_N_saw], par_t, sizeof par_t);
	integrate_table(g_pilut[SAU_WAVE_N_hsi], hsi_t, vs);
	integrate_table(g_pilut[SAU_WAVE_N_spa], spa_t, vs);
}

static void ensure_tables(void) {
	if (!g_tables_ready) {
		build_tables();
		g_tables_ready = 1;
	}
}

/** Replace the twelve PILUTs (12 x 2048 f32, wave-id order). */
ORA_API void ora_set_piluts(const float *tables) {
	memcpy(g_pilut, tables, sizeof g_pilut);
	g_tables_ready = 1;
}
ORA_API const float *ora_get_piluts(void) {
	ensure_tables();
	return &g_pilut[0][0];
}

/* sau/wave.h:127-141 -- 4-point 3rd-order Hermite, f32 taps, f64 polynomial */
static inline double herp(const float *lut, uint32_t phase) {
	uint32_t ind = phase >> SLEN_BITS;
	float s0 = lut[(ind - 1) & WMASK];
	float s1 = lut[ind];
	float s2 = lut[(ind + 1) & WMASK];
	float s3 = lut[(ind + 2) & WMASK];
	double x = ((phase & ((1u << SLEN_BITS) - 1)) * (1.f / (1u << SLEN_BITS)));
	double c0 = s1;
	double c1 = 1 / 2.0 * (s2 - s0);
	double c2 = s0 - 5 / 2.0 * s1 + 2 * s2 - 1 / 2.0 * s3;
	double c3 = 1 / 2.0 * (s3 - s0) + 3 / 2.0 * (s1 - s2);
	return ((c3 * x + c2) * x + c1) * x + c0;
}
/* ... its polynomial part alone (herp() = herp_rise() + the table value lut[phase >> SLEN_BITS]) */
static inline double herp_rise(const float *lut, uint32_t phase) {
	uint32_t ind = phase >> SLEN_BITS;
	float s0 = lut[(ind - 1) & WMASK];
	float s1 = lut[ind];
	float s2 = lut[(ind + 1) & WMASK];
	float s3 = lut[(ind + 2) & WMASK];
	double x = ((phase & ((1u << SLEN_BITS) - 1)) * (1.f / (1u << SLEN_BITS)));
	double c1 = 1 / 2.0 * (s2 - s0);
	double c2 = s0 - 5 / 2.0 * s1 + 2 * s2 - 1 / 2.0 * s3;
	double c3 = 1 / 2.0 * (s3 - s0) + 3 / 2.0 * (s1 - s2);
	return ((c3 * x + c2) * x + c1) * x;
}

ORA_API double ora_herp(int wave, uint32_t phase) {
	ensure_tables();
	return herp(g_pilut[wave], phase);
}

/* sau/wave.h:144-149 */
static inline float dvscale(int wave) {
	return g_picoeffs[wave].amp_scale * 0.125f * (float)UINT32_MAX;
}
static inline float dvoffset(int wave) { return g_picoeffs[wave].amp_dc; }

/* ---------------------------------------------------------------------- */
/* ramps (sau/line.c, sau/line.h)                                          */
/* ---------------------------------------------------------------------- */

/* sau/line.h:174-183 */
static inline float sinramp(float x) {
	const float k0 = +1.5702137061703461473139223358864L;
	const float k1 = -2.568278787380814155456160152724L;
	const float k2 = +1.1496958507977182668618673644367L;
	float x2 = x * x;
	return x * (k0 + x2 * (k1 + x2 * k2));
}
/* sau/line.h:195-200 */
static int g_fm_forms = 2; /* 0 strict, 1 main-loop forms, 2 = 1 + loop-tail forms */
static inline float expramp6(float x) {
	float x2 = x * x;
	float x3 = x2 * x;
	if (g_fm_forms) /* gcc -ffast-math factors x2 out: x3 + ((x*c1 + x2*c2)*(x3 - 1))*x2 */
		return x3 + ((x * (629.f / 1792.f) + x2 * (1163.f / 1792.f)) * (x3 + -1.f)) * x2;
	return x3 + (x2 * x3 - x2) * (x * (629.f / 1792.f) + x2 * (1163.f / 1792.f));
}

/* Scalar shape functions, sau/line.h:153-266 (used by the R oscillator's
 * segment mapping and by its self-modulating variant). */
static float shape_val(int type, float x, float a, float b) {
	switch (type) {
	default:
	case SAU_LINE_N_sah: return a;
	case SAU_LINE_N_lin: return a + (b - a) * x;
	case SAU_LINE_N_cos: return a + (b - a) * (sinramp(x - 0.5f) + 0.5f);
	case SAU_LINE_N_exp:
		return (a > b) ? b + (a - b) * expramp6(1.f - x)
		               : a + (b - a) * expramp6(x);
	case SAU_LINE_N_log:
		return (a < b) ? b + (a - b) * expramp6(1.f - x)
		               : a + (b - a) * expramp6(x);
	case SAU_LINE_N_xpe: return b + (a - b) * expramp6(1.f - x);
	case SAU_LINE_N_lge: return a + (b - a) * expramp6(x);
	case SAU_LINE_N_sqe: { float y = 1.f - x; return b + (a - b) * (y * y); }
	case SAU_LINE_N_cub: {
		float y = (0.5f - x) * 2;
		if (g_fm_forms)
			return b + (y * y * y + 1.f) * ((a - b) * 0.5f);
		return b + (a - b) * (y * y * y * 0.5f + 0.5f);
	}
	case SAU_LINE_N_smo:
		if (g_fm_forms)
			return a + (((b - a) * x) * (x * x)) * ((x * 6.f + -15.f) * x + 10.f);
		return a + (b - a) * x * x * x * (+10.f + x * (-15.f + x * +6.f));
	case SAU_LINE_N_uwh: {
		union { float f; int32_t i; } xs = {.f = x};
		int32_t s = ranfast32(xs.i);
		return a + (b - a) * (0.5f + (0.5f * 0x1p-31f) * s);
	}
	case SAU_LINE_N_ncl: {
		union { float f; int32_t i; } xs = {.f = x};
		int32_t s = ranfast32(xs.i);
		if (g_fm_forms) {
			float t = ((x + x) + -3.f) * x + 1.f;
			return a + (b - a) * ((t * (float)s) * (x * (0.5f * 0x1p-31f)) + x);
		}
		float xb = x; xb -= (3.f - (xb + xb)) * xb * xb;
		return a + (b - a) * (x + xb * s * (0.5f * 0x1p-31f));
	}
	case SAU_LINE_N_nhl: {
		union { float f; int32_t i; } xs = {.f = x};
		int32_t s = ranfast32(xs.i);
		if (g_fm_forms)
			return a + (b - a) * (((float)s * (1.f - x)) * (x * 0x1p-31f) + x);
		float xb = x; xb -= xb * xb;
		return a + (b - a) * (x + xb * s * 0x1p-31f);
	}
	}
}

/* Select between strict C order (0) and the order the main (4-wide) loops of
 * the reference's gcc -O3 -ffast-math build of sau/line.c execute (1, default).
 * The forms were read off the disassembly of oracle/_ref/line.o; the scalar
 * loop tails of that build use yet another association for `cub`, which is
 * not reproduced (it would tie the result to the reference's chunk lengths). */
ORA_API void ora_set_fastmath_forms(int on) { g_fm_forms = on; }

/* One ramp value at index k = i + pos of a sweep of `time` samples.
 * sau/line.c:27-37 (generic form), 65-281 (specialised forms). */
static void ramp_fill(int type, float *buf, uint32_t len, float v0, float vt,
		uint32_t pos, uint32_t time, const float *mulbuf) {
	const float inv_time = 1.f / time;
	const int32_t adj_pos = pos - (time / 2);
	switch (type) {
	default:
	case SAU_LINE_N_sah:
		for (uint32_t i = 0; i < len; ++i)
			buf[i] = mulbuf ? (v0 * mulbuf[i]) : v0;
		return;
	case SAU_LINE_N_lin: {
		const float vm = (v0 + vt) * 0.5f;
		const float vd = (vt - v0);
		if (g_fm_forms) {
			/* gcc hoists vd*inv_time out of the loop */
			const float k = vd * inv_time;
			for (uint32_t i = 0; i < len; ++i) {
				float v = vm + k * (float)((int32_t)i + adj_pos);
				buf[i] = mulbuf ? (v * mulbuf[i]) : v;
			}
		} else {
			for (uint32_t i = 0; i < len; ++i) {
				float x = ((int32_t)i + adj_pos) * inv_time;
				float v = vm + vd * x;
				buf[i] = mulbuf ? (v * mulbuf[i]) : v;
			}
		}
		return;
	}
	case SAU_LINE_N_cos: {
		const float vm = (v0 + vt) * 0.5f;
		const float vd = (vt - v0);
		for (uint32_t i = 0; i < len; ++i) {
			float x = ((int32_t)i + adj_pos) * inv_time;
			float v;
			if (g_fm_forms) { /* (vd*x) * poly(x*x) */
				float x2 = x * x;
				v = vm + (vd * x) * ((x2 * +1.1496958507977182668618673644367f
					+ -2.568278787380814155456160152724f) * x2
					+ +1.5702137061703461473139223358864f);
			} else
				v = vm + vd * sinramp(x);
			buf[i] = mulbuf ? (v * mulbuf[i]) : v;
		}
		return;
	}
	case SAU_LINE_N_exp:
		ramp_fill(v0 > vt ? SAU_LINE_N_xpe : SAU_LINE_N_lge,
				buf, len, v0, vt, pos, time, mulbuf);
		return;
	case SAU_LINE_N_log:
		ramp_fill(v0 < vt ? SAU_LINE_N_xpe : SAU_LINE_N_lge,
				buf, len, v0, vt, pos, time, mulbuf);
		return;
	case SAU_LINE_N_xpe:
	case SAU_LINE_N_lge:
	case SAU_LINE_N_smo:
		for (uint32_t i = 0; i < len; ++i) {
			float x = (i + pos) * inv_time;
			float v = shape_val(type, x, v0, vt);
			buf[i] = mulbuf ? (v * mulbuf[i]) : v;
		}
		return;
	case SAU_LINE_N_sqe:
		for (uint32_t i = 0; i < len; ++i) {
			float x = 0.5f - ((int32_t)i + adj_pos) * inv_time;
			float v = vt + (v0 - vt) * (x * x);
			buf[i] = mulbuf ? (v * mulbuf[i]) : v;
		}
		return;
	case SAU_LINE_N_cub: {
		const float scale = -2 * inv_time;
		for (uint32_t i = 0; i < len; ++i) {
			float x = ((int32_t)i + adj_pos) * scale;
			float h = (v0 - vt) * 0.5f;
			float v;
			if (g_fm_forms == 2 && (len & 1) && i == len - 1)
				v = vt + (x * x * x * h + h); /* gcc's scalar loop tail */
			else if (g_fm_forms)
				v = vt + (x * x * x + 1.f) * h;
			else
				v = vt + (v0 - vt) * (x * x * x * 0.5f + 0.5f);
			buf[i] = mulbuf ? (v * mulbuf[i]) : v;
		}
		return;
	}
	case SAU_LINE_N_uwh: {
		const float scale = 0.5f / (float)INT32_MAX;
		const float vm = (v0 + vt) * 0.5f;
		const float vd = (vt - v0) * scale;
		for (uint32_t i = 0; i < len; ++i) {
			int32_t s = ranfast32(pos + i);
			float v = vm + vd * s;
			buf[i] = mulbuf ? (v * mulbuf[i]) : v;
		}
		return;
	}
	case SAU_LINE_N_ncl: {
		const float scale = 0.5f / (float)INT32_MAX;
		const float vm = (v0 + vt) * 0.5f;
		const float vd = (vt - v0);
		for (uint32_t i = 0; i < len; ++i) {
			float x = ((int32_t)i + adj_pos) * inv_time;
			int32_t s = ranfast32(pos + i);
			float v;
			if (g_fm_forms) {
				float xb0 = x + 0.5f;
				float t = ((xb0 + xb0) + -3.f) * xb0 + 1.f;
				v = vm + vd * (((float)s * t) * (xb0 * scale) + x);
			} else {
				float xb = x + 0.5f; xb -= (3.f - (xb + xb)) * xb * xb;
				v = vm + vd * (x + xb * s * scale);
			}
			buf[i] = mulbuf ? (v * mulbuf[i]) : v;
		}
		return;
	}
	case SAU_LINE_N_nhl: {
		const float scale = 2 * 0.5f / (float)INT32_MAX;
		const float vm = (v0 + vt) * 0.5f;
		const float vd = (vt - v0);
		for (uint32_t i = 0; i < len; ++i) {
			float x = ((int32_t)i + adj_pos) * inv_time;
			int32_t s = ranfast32(pos + i);
			float v;
			if (g_fm_forms) {
				float xb0 = x + 0.5f;
				v = vm + vd * (((float)s * (1.f - xb0)) * (xb0 * scale) + x);
			} else {
				float xb = x + 0.5f; xb -= xb * xb;
				v = vm + vd * (x + xb * s * scale);
			}
			buf[i] = mulbuf ? (v * mulbuf[i]) : v;
		}
		return;
	}
	}
}

ORA_API void ora_ramp_fill(int type, float *buf, uint32_t len, float v0,
		float vt, uint32_t pos, uint32_t time, const float *mulbuf) {
	ramp_fill(type, buf, len, v0, vt, pos, time, mulbuf);
}

/* sau/line.c:16-24 */
static void ramp_map(int type, float *buf, uint32_t len,
		const float *end0, const float *end1) {
	uint32_t n = len;
	if (type == SAU_LINE_N_cub && g_fm_forms == 2) {
		/* gcc's 2-wide and scalar loop tails: y^3*h + h */
		n = len & ~3u;
		for (uint32_t i = n; i < len; ++i) {
			float y = (0.5f - buf[i]) * 2;
			float h = (end0[i] - end1[i]) * 0.5f;
			buf[i] = end1[i] + (y * y * y * h + h);
		}
	}
	for (uint32_t i = 0; i < n; ++i)
		buf[i] = shape_val(type, buf[i], end0[i], end1[i]);
}
ORA_API void ora_ramp_map(int type, float *buf, uint32_t len,
		const float *end0, const float *end1) {
	ramp_map(type, buf, len, end0, end1);
}

/* Ramp state; same fields as the host's sauLine. */
typedef sauLine Ramp;

/* sau/line.c:349-378 */
static uint32_t ramp_get(Ramp *o, float *buf, uint32_t buf_len, const float *mulbuf) {
	if (!(o->flags & SAU_LINEP_GOAL))
		return 0;
	if (o->flags & SAU_LINEP_GOAL_RATIO) {
		if (!(o->flags & SAU_LINEP_STATE_RATIO)) {
			if (mulbuf) o->v0 /= mulbuf[0];
			o->flags |= SAU_LINEP_STATE_RATIO;
		}
	} else {
		if (o->flags & SAU_LINEP_STATE_RATIO) {
			if (mulbuf) o->v0 *= mulbuf[0];
			o->flags &= ~SAU_LINEP_STATE_RATIO;
		}
		mulbuf = NULL;
	}
	if (o->pos >= o->end)
		return 0;
	uint32_t len = o->end - o->pos;
	if (len > buf_len) len = buf_len;
	ramp_fill(o->type, buf, len, o->v0, o->vt, o->pos, o->end, mulbuf);
	return len;
}

/* sau/line.c:385-398 */
static bool ramp_advance(Ramp *o, uint32_t buf_len) {
	if (o->pos < o->end) {
		uint32_t len = o->end - o->pos;
		if (len > buf_len) len = buf_len;
		o->pos += len;
	}
	if (o->pos >= o->end) {
		o->pos = 0;
		o->flags &= ~SAU_LINEP_TIME;
		return false;
	}
	return true;
}

/* sau/line.c:417-445 */
static void ramp_run(Ramp *o, float *buf, uint32_t buf_len, const float *mulbuf) {
	uint32_t len = 0;
	bool hold;
	if (!(o->flags & SAU_LINEP_GOAL)) {
		ramp_advance(o, buf_len);
		hold = true;
	} else {
		len = ramp_get(o, buf, buf_len, mulbuf);
		o->pos += len;
		hold = (o->pos >= o->end);
		if (hold) {
			o->v0 = o->vt;
			o->pos = 0;
			o->flags &= ~(SAU_LINEP_GOAL | SAU_LINEP_GOAL_RATIO | SAU_LINEP_TIME);
		}
	}
	if (hold) {
		if (!(o->flags & SAU_LINEP_STATE_RATIO))
			mulbuf = NULL;
		else if (mulbuf)
			mulbuf += len;
		ramp_fill(SAU_LINE_N_sah, buf + len, buf_len - len, o->v0, o->v0, 0, 0, mulbuf);
	}
}

/* sau/line.c:456-473 */
static void ramp_skip(Ramp *o, uint32_t skip_len) {
	if (!ramp_advance(o, skip_len)) {
		if (!(o->flags & SAU_LINEP_GOAL))
			return;
		o->v0 = o->vt;
		if (o->flags & SAU_LINEP_GOAL_RATIO)
			o->flags |= SAU_LINEP_STATE_RATIO;
		else
			o->flags &= ~SAU_LINEP_STATE_RATIO;
		o->flags &= ~(SAU_LINEP_GOAL | SAU_LINEP_GOAL_RATIO);
	}
}

/* sau/line.c:287-332 */
static void ramp_copy(Ramp *o, const sauLine *src, uint32_t srate) {
	if (!src)
		return;
	uint8_t mask = 0;
	if (src->flags & SAU_LINEP_STATE) {
		o->v0 = src->v0;
		mask |= SAU_LINEP_STATE | SAU_LINEP_STATE_RATIO;
	} else if (o->flags & SAU_LINEP_GOAL) {
		if (src->flags & SAU_LINEP_GOAL) {
			float f;
			ramp_get(o, &f, 1, NULL);
			o->v0 = f;
		}
	}
	if (src->flags & SAU_LINEP_GOAL) {
		o->vt = src->vt;
		if (src->flags & SAU_LINEP_TIME_IF_NEW)
			o->end -= o->pos;
		o->pos = 0;
		mask |= SAU_LINEP_GOAL | SAU_LINEP_GOAL_RATIO;
	}
	if (src->flags & SAU_LINEP_TYPE) {
		o->type = src->type;
		mask |= SAU_LINEP_TYPE;
	}
	if (!(o->flags & SAU_LINEP_TIME) || !(src->flags & SAU_LINEP_TIME_IF_NEW)) {
		if (src->flags & SAU_LINEP_TIME) {
			o->end = ms_to_samples(src->time_ms, srate, NULL);
			o->time_ms = src->time_ms;
			mask |= SAU_LINEP_TIME;
		}
	}
	o->flags &= ~mask;
	o->flags |= (src->flags & mask);
}

ORA_API void ora_ramp_copy(sauLine *dst, const sauLine *src, uint32_t srate) { ramp_copy(dst, src, srate); }
ORA_API void ora_ramp_run(sauLine *o, float *buf, uint32_t len, const float *mulbuf) { ramp_run(o, buf, len, mulbuf); }
ORA_API void ora_ramp_skip(sauLine *o, uint32_t len) { ramp_skip(o, len); }

/* ---------------------------------------------------------------------- */
/* noise (sau/generator/noise.h)                                           */
/* ---------------------------------------------------------------------- */

/* noise.h:61-70 */
static inline float soft_sqrtm2logp1(float x) {
	const float k0 = -0.80270565422983103084;
	const float k1 = +5.52274428214641442648;
	const float k2 = -138.87126103150588693697;
	float x2 = x * x;
	float x4 = x2 * x2;
	return 0.5f + x * (k0 + x4 * (k1 + x4 * k2));
}
/* noise.h:77-81 */
static inline float ssgauss_dist4(float x) {
	float x2 = x * x;
	float gx = (x + x2) * 0.5f;
	return x * (1 - gx * (1 - x2));
}
/* noise.h:90-98 */
static inline float franssgauss32(uint32_t n) {
	int32_t s0 = ranfast32(n);
	int32_t s1 = mcg32(s0);
	float a = s0 * 0x1p-32;
	float b = s1 * 0x1p-32;
	float c = ssgauss_dist4(soft_sqrtm2logp1(a));
	b = c * sinpi_d5f(b);
	return b;
}
ORA_API float ora_franssgauss32(uint32_t n) { return franssgauss32(n); }
ORA_API uint32_t ora_ranfast32(uint32_t n) { return ranfast32(n); }

typedef struct Noise { uint32_t n, prev; uint8_t type; } Noise;

/* noise.h:41-185 */
static void noise_run(Noise *o, float *buf, size_t len) {
	const float scale = 0x1p-31;
	switch (o->type) {
	default:
	case SAU_NOISE_N_wh:
		for (size_t i = 0; i < len; ++i)
			buf[i] = fscalei(ranfast32(o->n++), scale);
		break;
	case SAU_NOISE_N_gw:
		for (size_t i = 0; i < len; ++i)
			buf[i] = franssgauss32(o->n++);
		break;
	case SAU_NOISE_N_bw:
		for (size_t i = 0; i < len; ++i) {
			uint32_t n = o->n++;
			int32_t s = sar32(ranfast32(n), 31) * 2 + 1;
			buf[i] = s;
		}
		break;
	case SAU_NOISE_N_tw:
		for (size_t i = 0; i < len; ++i) {
			uint32_t n = o->n++;
			int32_t s = sar32(ranfast32(n), 31) * 2 + 1;
			buf[i] = (n & 1) ? s : 0.f;
		}
		break;
	case SAU_NOISE_N_re: {
		uint32_t sum = o->prev;
		for (size_t i = 0; i < len; ++i) {
			int32_t s = ranfast32(o->n++);
			sum += (s >> 6);
			s = foldhd32(sum);
			buf[i] = fscalei(s, scale);
		}
		o->prev = sum;
		break;
	}
	case SAU_NOISE_N_vi: {
		uint32_t s0 = o->prev;
		for (size_t i = 0; i < len; ++i) {
			uint32_t s1 = ranfast32(o->n++);
			buf[i] = fscalei((s1 / 2) - (s0 / 2), scale);
			s0 = s1;
		}
		o->prev = s0;
		break;
	}
	case SAU_NOISE_N_bv: {
		int32_t s0 = o->prev;
		for (size_t i = 0; i < len; ++i) {
			uint32_t n = o->n++;
			int32_t s1 = sar32(ranfast32(n), 31);
			s1 = (n & 1) ? (s1 * 2 + 1) : 0;
			buf[i] = (s1 - s0);
			s0 = s1;
		}
		o->prev = s0;
		break;
	}
	}
}

/* ---------------------------------------------------------------------- */
/* operator state                                                          */
/* ---------------------------------------------------------------------- */

#define BLOCK_MAX 1024 /* generator.c:28 */
#define SLOTS_PER_LEVEL 7 /* generator.c:133 */

enum { OPF_INIT = 1, OPF_VISITED = 2, OPF_TIME_INF = 4 }; /* generator.c:39-43 */
enum { OSC_RESET_DIFF = 1, OSC_RESET = 1 };                /* wosc.h:37-38 */

typedef struct RasState { /* rasg.h:29-39 */
	uint64_t cycle_phase;
	bool rate2x;
	uint8_t line;
	unsigned flags, func, level;
	uint32_t alpha;
} RasState;

typedef struct Op {
	uint32_t time;
	uint8_t kind, flags;
	Ramp amp, amp2, pan, freq, freq2, pma;
	const sauProgramIDArr *camods, *amods, *ramods, *fmods, *rfmods,
	                      *pmods, *apmods, *fpmods;
	float coeff;            /* 2^32 / srate, wosc.h:30, rasg.h:27 */
	/* W */
	uint32_t phase, prev_phase;
	double prev_Is;
	uint8_t wave, osc_flags;
	/* W and R feedback */
	float prev_s, fb_s;
	/* R */
	RasState ras;
	/* N */
	Noise noise;
} Op;

typedef struct Voice { /* generator.c:97-102 */
	uint32_t duration;
	uint8_t flags, freq_buf_id;
	uint32_t carr_op_id;
} Voice;

typedef struct Event { uint32_t wait; const sauProgramEvent *pe; } Event;

typedef float Buf[BLOCK_MAX];

typedef struct OraGen {
	uint32_t srate;
	bool out_dirty_cleared;
	uint32_t mix_used;
	Buf *bufs, *mix;
	size_t event, ev_count;
	Event *events;
	uint32_t event_pos;
	uint32_t voice, vo_count;
	Voice *voices;
	float amp_scale;
	uint32_t op_count;
	Op *ops;
	uint32_t block_len;
} OraGen;

static const sauProgramIDArr g_no_ids = {0};

/* generator.c:135-217 */
ORA_API OraGen *ora_create(const sauProgram *prg, uint32_t srate) {
	ensure_tables();
	OraGen *o = calloc(1, sizeof *o);
	if (!o) return NULL;
	o->srate = srate;
	o->block_len = BLOCK_MAX;
	o->ev_count = prg->ev_count;
	o->vo_count = prg->vo_count;
	o->op_count = prg->op_count;
	o->events = calloc(o->ev_count ? o->ev_count : 1, sizeof(Event));
	o->voices = calloc(o->vo_count ? o->vo_count : 1, sizeof(Voice));
	o->ops = calloc(o->op_count ? o->op_count : 1, sizeof(Op));
	o->bufs = calloc((1 + prg->op_nest_depth) * SLOTS_PER_LEVEL, sizeof(Buf));
	o->mix = calloc(2, sizeof(Buf));
	if (!o->events || !o->voices || !o->ops || !o->bufs || !o->mix) return NULL;
	o->amp_scale = 0.5f * prg->ampmult;
	if (prg->mode & SAU_PMODE_AMP_DIV_VOICES)
		o->amp_scale /= o->vo_count;
	int carry = 0;
	for (size_t i = 0; i < prg->ev_count; ++i) {
		o->events[i].wait = ms_to_samples(prg->events[i].wait_ms, srate, &carry);
		o->events[i].pe = &prg->events[i];
	}
	return o;
}

ORA_API void ora_destroy(OraGen *o) {
	if (!o) return;
	free(o->events); free(o->voices); free(o->ops); free(o->bufs); free(o->mix);
	free(o);
}

/** Internal block length (<= 1024); the output must not depend on it. */
ORA_API void ora_set_block_len(OraGen *o, uint32_t n) {
	if (n >= 1 && n <= BLOCK_MAX) o->block_len = n;
}

/* ---- R oscillator option/phase setters, rasg.h:59-119 ------------------ */

static uint32_t ras_get_cycle(const RasState *r) {
	return (uint32_t)(r->cycle_phase >> 32) & ~1u;
}
static uint32_t ras_get_phase(const RasState *r) {
	return r->rate2x ? (uint32_t)(r->cycle_phase >> 1) : (uint32_t)r->cycle_phase;
}
static void ras_set_cycle(RasState *r, uint32_t cycle) {
	uint32_t phase = ras_get_phase(r);
	uint64_t p64 = r->rate2x ? ((uint64_t)phase) << 1 : phase;
	r->cycle_phase = ((uint64_t)(cycle & ~1u)) << 32 | p64;
}
static void ras_set_phase(RasState *r, uint32_t phase) {
	uint32_t cycle = ras_get_cycle(r);
	uint64_t p64 = r->rate2x ? ((uint64_t)phase) << 1 : phase;
	r->cycle_phase = ((uint64_t)cycle) << 32 | p64;
}
static void ras_set_opt(RasState *r, const sauRasOpt *opt) {
	unsigned flags = opt->flags;
	if (opt->flags & SAU_RAS_O_LINE_SET) r->line = opt->line;
	if (opt->flags & SAU_RAS_O_FUNC_SET) r->func = opt->func;
	else flags |= r->flags;
	if (opt->flags & SAU_RAS_O_LEVEL_SET) r->level = opt->level;
	if (opt->flags & SAU_RAS_O_ASUBVAL_SET) r->alpha = opt->alpha;
	r->flags = flags & 0x3ff; /* 10-bit field in the reference */
	bool rate2x = !(flags & SAU_RAS_O_HALFSHAPE);
	if (rate2x != r->rate2x) {
		uint32_t cycle = ras_get_cycle(r);
		uint32_t phase = ras_get_phase(r);
		r->rate2x = rate2x;
		ras_set_cycle(r, cycle);
		ras_set_phase(r, phase);
	}
}
/* program.h:146-148 */
static unsigned ras_level(unsigned digit) {
	return digit <= 6 ? digit : (digit - 4) * (digit - 4) + 2;
}

/* generator.c:245-278 with wosc.h:55-71, rasg.h:44-57 */
static void op_prepare(OraGen *o, Op *n, Voice *vn, const sauProgramOpData *od) {
	if (od->use_type == SAU_POP_N_carr)
		vn->freq_buf_id = 0;
	memset(n, 0, sizeof *n);
	n->coeff = (float)(0x1p32 / o->srate);
	switch (od->type) {
	case SAU_POPT_N_wave:
		n->phase = g_picoeffs[SAU_WAVE_N_sin].phase_adj;
		n->wave = SAU_WAVE_N_sin;
		n->osc_flags = OSC_RESET;
		if (od->use_type == SAU_POP_N_carr) vn->freq_buf_id = 3 - 1;
		break;
	case SAU_POPT_N_raseg:
		n->ras.cycle_phase = 0;
		n->ras.rate2x = true;
		n->ras.line = SAU_LINE_N_lin;
		n->ras.func = SAU_RAS_F_URAND;
		n->ras.level = ras_level(9);
		n->ras.alpha = 0x9e3779b9u;
		n->ras.flags = 0;
		if (od->use_type == SAU_POP_N_carr) vn->freq_buf_id = 4 - 1;
		break;
	default: break;
	}
	n->camods = n->amods = n->ramods = n->fmods = n->rfmods =
		n->pmods = n->apmods = n->fpmods = &g_no_ids;
	n->kind = od->type;
	n->flags = OPF_INIT;
}

/* generator.c:283-343 with wosc.h:73-91, noise.h:29-36 */
static void op_update(OraGen *o, Op *n, const sauProgramOpData *od) {
	uint32_t params = od->params;
	bool osc = false;
	switch (od->type) {
	case SAU_POPT_N_noise:
		if (params & SAU_POPP_MODE) { n->noise.type = od->mode.main; n->noise.prev = 0; }
		if (params & SAU_POPP_SEED) n->noise.n = od->seed;
		break;
	case SAU_POPT_N_wave:
		if (params & SAU_POPP_MODE) {
			uint8_t wave = od->mode.main;
			uint32_t old_adj = g_picoeffs[n->wave].phase_adj;
			uint32_t adj = g_picoeffs[wave].phase_adj;
			n->phase += adj - old_adj;
			n->wave = wave;
			n->osc_flags |= OSC_RESET_DIFF;
		}
		if (params & SAU_POPP_PHASE)
			n->phase = od->phase + g_picoeffs[n->wave].phase_adj;
		osc = true;
		break;
	case SAU_POPT_N_raseg:
		if (params & SAU_POPP_MODE) ras_set_opt(&n->ras, &od->mode.ras);
		if (params & SAU_POPP_PHASE) ras_set_phase(&n->ras, od->phase);
		if (params & SAU_POPP_SEED) ras_set_cycle(&n->ras, od->seed);
		osc = true;
		break;
	default: break;
	}
	if (osc) {
		if (od->fmods) n->fmods = od->fmods;
		if (od->rfmods) n->rfmods = od->rfmods;
		if (od->pmods) n->pmods = od->pmods;
		if (od->apmods) n->apmods = od->apmods;
		if (od->fpmods) n->fpmods = od->fpmods;
		ramp_copy(&n->freq, od->freq, o->srate);
		ramp_copy(&n->freq2, od->freq2, o->srate);
		ramp_copy(&n->pma, od->pm_a, o->srate);
	}
	if (params & SAU_POPP_TIME) {
		if (od->time.flags & SAU_TIMEP_IMPLICIT) {
			n->time = 0;
			n->flags |= OPF_TIME_INF;
		} else {
			n->time = ms_to_samples(od->time.v_ms, o->srate, NULL);
			n->flags &= ~OPF_TIME_INF;
		}
	}
	if (od->camods) n->camods = od->camods;
	if (od->amods) n->amods = od->amods;
	if (od->ramods) n->ramods = od->ramods;
	ramp_copy(&n->amp, od->amp, o->srate);
	ramp_copy(&n->amp2, od->amp2, o->srate);
	ramp_copy(&n->pan, od->pan, o->srate);
}

/* generator.c:348-377 */
static void apply_event(OraGen *o, const Event *e) {
	const sauProgramEvent *pe = e->pe;
	Voice *vn = NULL;
	if (pe->vo_id != SAU_PVO_NO_ID)
		vn = &o->voices[pe->vo_id];
	for (size_t i = 0; i < pe->op_data_count; ++i) {
		const sauProgramOpData *od = &pe->op_data[i];
		Op *n = &o->ops[od->id];
		if (!(n->flags & OPF_INIT))
			op_prepare(o, n, vn, od);
		op_update(o, n, od);
	}
	if (vn) {
		vn->carr_op_id = pe->carr_op_id;
		vn->flags |= 1;
		if (o->voice > pe->vo_id)
			o->voice = pe->vo_id;
		/* generator.c:233-240 */
		vn->duration = o->ops[vn->carr_op_id].time;
	}
}

/* ---------------------------------------------------------------------- */
/* oscillators                                                             */
/* ---------------------------------------------------------------------- */

/* wosc.h:135-169: pre-incremented u32 phase plus PM offsets */
static void phasor_fill(Op *n, uint32_t *phase_out, size_t len,
		const float *freq, const float *pm, const float *fpm) {
	const float fpm_scale = 1.f / 632.45553203367586639978;
	for (size_t i = 0; i < len; ++i) {
		float f = freq[i];
		uint32_t ofs = 0;
		if (pm && fpm) {
			float p = g_fm_forms ? pm[i] + ((fpm[i] * f) * fpm_scale)
			                     : pm[i] + (fpm[i] * fpm_scale * f);
			ofs = rint_i64(p * 0x1p31f);
		} else if (pm) {
			ofs = rint_i64(pm[i] * 0x1p31f);
		} else if (fpm) {
			/* the reference build folds the two constants (gcc -ffast-math) */
			if (g_fm_forms)
				ofs = rint_i64((fpm[i] * f) * (fpm_scale * 0x1p31f));
			else {
				float p = fpm[i] * fpm_scale * f;
				ofs = rint_i64(p * 0x1p31f);
			}
		}
		uint32_t inc = rint_i64(n->coeff * f);
		phase_out[i] = ofs + (n->phase += inc);
	}
}

/* wosc.h:215-231 */
static void wosc_reset(Op *n, uint32_t phase) {
	const float *lut = g_pilut[n->wave];
	const float diff_scale = dvscale(n->wave);
	const float diff_offset = dvoffset(n->wave);
	if (n->osc_flags & OSC_RESET_DIFF) {
		int32_t phase_diff = 1 << SLEN_BITS;
		n->prev_Is = herp(lut, phase - phase_diff);
		double Is = herp(lut, phase);
		double x = (diff_scale / phase_diff);
		if (g_fm_forms) {
			/* the reference build: sauWOsc_reset is a function of its own there, and gcc's fast-math subtracts the
			 * table value and the polynomial part of the earlier Hermite value one after the other,
			 * (Is - y1') - P', not their rounded sum (objdump of oracle/_ref/generator.o; pinned by
			 * tests/test_oracle.py: the `ean` feedback case) */
			const uint32_t pp = phase - (uint32_t)phase_diff;
			n->prev_s = ((Is - (double)lut[pp >> SLEN_BITS]) - herp_rise(lut, pp)) * x + diff_offset;
		} else
		n->prev_s = (Is - n->prev_Is) * x + diff_offset;
		n->prev_Is = Is;
		n->prev_phase = phase;
	}
	n->osc_flags &= ~OSC_RESET;
}

/* wosc.h:238-266 and 273-310 (pm_a != NULL) */
static void wosc_run(Op *n, float *out, size_t len, const uint32_t *phase_buf,
		const float *pm_a) {
	const float *lut = g_pilut[n->wave];
	const float diff_scale = dvscale(n->wave);
	const float diff_offset = dvoffset(n->wave);
	if (len > 0 && (n->osc_flags & OSC_RESET))
		wosc_reset(n, phase_buf[0]);
	for (size_t i = 0; i < len; ++i) {
		float s;
		uint32_t phase = phase_buf[i];
		if (pm_a)
			phase += (uint32_t)rint_i64(n->fb_s * pm_a[i] * 0x1p31f);
		int32_t phase_diff = phase - n->prev_phase;
		if (phase_diff == 0) {
			s = n->prev_s;
		} else {
			double Is = herp(lut, phase);
			double x = (diff_scale / phase_diff);
			s = (Is - n->prev_Is) * x + diff_offset;
			n->prev_Is = Is;
			n->prev_s = s;
			n->prev_phase = phase;
		}
		out[i] = s;
		if (pm_a)
			n->fb_s = (n->fb_s + s) * 0.5f;
	}
}

/* rasg.h:165-222: post-incremented u64 cycle|phase */
static void cyclor_fill(Op *n, uint32_t *cycle_out, float *phase_out, size_t len,
		const float *freq, const float *pm, const float *fpm) {
	const float fpm_scale = 1.f / 632.45553203367586639978;
	float coeff = n->coeff;
	float phase_scale = 0x1p31f;
	if (n->ras.rate2x) {
		coeff *= 2;
		phase_scale *= 2;
	}
	for (size_t i = 0; i < len; ++i) {
		float f = freq[i];
		int64_t ofs = 0;
		if (pm && fpm) {
			float p = g_fm_forms ? pm[i] + ((fpm[i] * f) * fpm_scale)
			                     : pm[i] + (fpm[i] * fpm_scale * f);
			ofs = rint_i64(p * phase_scale);
		} else if (pm) {
			ofs = rint_i64(pm[i] * phase_scale);
		} else if (fpm) {
			if (g_fm_forms)
				ofs = rint_i64((fpm[i] * f) * (fpm_scale * phase_scale));
			else {
				float p = fpm[i] * fpm_scale * f;
				ofs = rint_i64(p * phase_scale);
			}
		}
		uint64_t cp = (uint64_t)ofs + n->ras.cycle_phase;
		n->ras.cycle_phase += (uint64_t)rint_i64(coeff * f);
		cycle_out[i] = cp >> 32;
		uint32_t ph = ((uint32_t)cp) >> 1;
		phase_out[i] = ((int32_t)ph) * 0x1p-31f;
	}
}

/* Segment end values for one cycle index; rasg.h:299-671. */
typedef struct RasMapCtx {
	unsigned func, flags, level;
	uint32_t alpha;
	float vbin_scale;
} RasMapCtx;

static void ras_map_ctx(const RasState *r, RasMapCtx *c) {
	c->func = r->func; c->flags = r->flags; c->level = r->level; c->alpha = r->alpha;
	/* rasg.h:398-402 */
	const float scale_diff = 1.f - (sar32(INT32_MAX, r->level) / 0x1p31f);
	c->vbin_scale = (1.f + scale_diff * scale_diff) / 0x1p31f;
}

static inline void ras_ends(const RasMapCtx *c, uint32_t cycle, float *pa, float *pb) {
	const int sr = c->level;
	const bool violet = (c->flags & SAU_RAS_O_VIOLET) != 0;
	float a, b;
	switch (c->func) {
	default:
	case SAU_RAS_F_URAND:
		if (violet) { /* rasg.h:307-313 */
			uint32_t s0 = ranfast32(cycle - 1) / 2;
			uint32_t s1 = ranfast32(cycle) / 2;
			uint32_t s2 = ranfast32(cycle + 1) / 2;
			a = fscalei(s1 - s0, 0x1p-31f);
			b = fscalei(s2 - s1, 0x1p-31f);
		} else { /* rasg.h:336-339 */
			a = fscalei(ranfast32(cycle), 0x1p-31f);
			b = fscalei(ranfast32(cycle + 1), 0x1p-31f);
		}
		break;
	case SAU_RAS_F_GAUSS: /* rasg.h:376-379 */
		a = franssgauss32(cycle);
		b = franssgauss32(cycle + 1);
		break;
	case SAU_RAS_F_BIN:
		if (violet) { /* rasg.h:405-416 */
			uint32_t sb = (cycle & 1) << 31;
			uint32_t sb_flip = (1u << 31) - sb;
			uint32_t s0 = divi(sar32(ranfast32(cycle - 1), sr) + sb, 2);
			uint32_t s1 = divi(sar32(ranfast32(cycle), sr) + sb_flip, 2);
			uint32_t s2 = divi(sar32(ranfast32(cycle + 1), sr) + sb, 2);
			a = fscalei(s1 - s0, c->vbin_scale);
			b = fscalei(s2 - s1, c->vbin_scale);
		} else { /* rasg.h:459-465 */
			uint32_t offs = INT32_MAX + (cycle & 1) * 2;
			uint32_t s1 = sar32(ranfast32(cycle), sr) + offs;
			uint32_t s2 = sar32(ranfast32(cycle + 1), sr) - offs;
			a = fscalei(s1, 0x1p-31f);
			b = fscalei(s2, 0x1p-31f);
		}
		break;
	case SAU_RAS_F_TERN: { /* rasg.h:509-517 */
		uint32_t sb = (cycle & 1) << 31;
		uint32_t sb_flip = (1u << 31) - sb;
		uint32_t s1 = sar32(ranfast32(cycle), sr) + sb_flip;
		uint32_t s2 = sar32(ranfast32(cycle + 1), sr) + sb;
		a = fscalei(s1, 0x1p-31f);
		b = fscalei(s2, 0x1p-31f);
		break;
	}
	case SAU_RAS_F_FIXED:
		if (c->level >= ras_level(9)) { /* rasg.h:538-541 */
			a = odd_sign(cycle);
			b = -a;
		} else if (violet) { /* rasg.h:563-576 */
			uint32_t sign = odd_sign(cycle);
			uint32_t s0 = divi(sign * ((ranfast32(cycle - 1) >> sr) - INT32_MAX), 2);
			uint32_t s1 = divi(-sign * ((ranfast32(cycle) >> sr) - INT32_MAX), 2);
			uint32_t s2 = divi(sign * ((ranfast32(cycle + 1) >> sr) - INT32_MAX), 2);
			a = fscalei(s1 - s0, 0x1p-31f);
			b = fscalei(s2 - s1, 0x1p-31f);
		} else { /* rasg.h:608-616 */
			uint32_t sign = odd_sign(cycle);
			a = fscalei(-sign * ((ranfast32(cycle) >> sr) - INT32_MAX), 0x1p-31f);
			b = fscalei(sign * ((ranfast32(cycle + 1) >> sr) - INT32_MAX), 0x1p-31f);
		}
		break;
	case SAU_RAS_F_ADDREC: { /* rasg.h:659-663 */
		uint32_t s0 = cycle * c->alpha;
		uint32_t s1 = (cycle + 1) * c->alpha;
		a = fscalei(s0, 0x1p-31f);
		b = fscalei(s1, 0x1p-31f);
		break;
	}
	}
	*pa = a; *pb = b;
}

/* sau/line.h:18-32 perlin_amp column */
static const float g_perlin_amp[SAU_LINE_NAMED] = {
	2.f, 2.f, 1.f, 1.55845810035f, 1.55845810035f, 1.55845810035f,
	1.55845810035f, 1.89339094650f, 2.f, 2.f, 2.f, 1.89339094650f, 1.f,
};

/* rasg.h:692-743 */
static void rasg_run(Op *n, size_t len, float *main_buf, float *end_a,
		float *end_b, const uint32_t *cycle_buf) {
	RasMapCtx c;
	ras_map_ctx(&n->ras, &c);
	for (size_t i = 0; i < len; ++i)
		ras_ends(&c, cycle_buf[i], &end_a[i], &end_b[i]);
	const unsigned flags = n->ras.flags, line = n->ras.line;
	if (flags & SAU_RAS_O_PERLIN) {
		const float perlin_amp =
			(flags & (SAU_RAS_O_HALFSHAPE | SAU_RAS_O_ZIGZAG)) ? 1.f : g_perlin_amp[line];
		for (size_t i = 0; i < len; ++i) {
			float phase = main_buf[i];
			if (g_fm_forms) {
				/* the reference build's block loop (vector body and scalar tail):
				 * (a * phase) * amp, (b * amp) * (phase - 1) */
				end_a[i] = (end_a[i] * phase) * perlin_amp;
				end_b[i] = (end_b[i] * perlin_amp) * (phase - 1.f);
			} else {
				end_a[i] *= perlin_amp * phase;
				end_b[i] *= perlin_amp * (phase - 1.f);
			}
		}
	}
	if (flags & SAU_RAS_O_HALFSHAPE) {
		for (size_t i = 0; i < len; ++i) {
			float a = end_a[i], b = end_b[i];
			end_a[i] = a < b ? b : a;
			end_b[i] = a > b ? b : a;
		}
	}
	if (flags & SAU_RAS_O_ZIGZAG) {
		float *t = end_a; end_a = end_b; end_b = t;
	}
	if (flags & SAU_RAS_O_SQUARE) {
		for (size_t i = 0; i < len; ++i) {
			end_a[i] *= fabsf(end_a[i]);
			end_b[i] *= fabsf(end_b[i]);
		}
	}
	ramp_map(line, main_buf, len, end_a, end_b);
}

/* rasg.h:242-280, 764-772: per-sample fused variant with feedback */
static void rasg_run_selfmod(Op *n, size_t len, float *main_buf,
		const uint32_t *cycle_buf, const float *pm_abuf) {
	RasMapCtx c;
	ras_map_ctx(&n->ras, &c);
	const unsigned flags = n->ras.flags, line = n->ras.line;
	const float perlin_amp =
		(flags & (SAU_RAS_O_HALFSHAPE | SAU_RAS_O_ZIGZAG)) ? 1.f : g_perlin_amp[line];
	for (size_t i = 0; i < len; ++i) {
		/* the reference build halves the amount first: fb_s * (0.5f * pm_abuf[i]) -- the same
		 * value unless a product leaves the normal range (generator.o, every sauRasG_map_*_s) */
		float pm_a = g_fm_forms ? n->fb_s * (0.5f * pm_abuf[i]) : n->fb_s * pm_abuf[i] * 0.5f;
		float phase = main_buf[i] + pm_a;
		int32_t cycle_adj;
		if (g_fm_forms) {
			/* -ffast-math inlines floorf: cvttss2si (0x80000000 out of range), then one less where the
			 * truncated value lies above phase -- in 32-bit integer arithmetic: below -2^31 the integer
			 * indefinite wraps to INT32_MAX (generator.o, every sauRasG_map_*_s) */
			cycle_adj = (fabsf(phase) < 0x1p31f) ? (int32_t)phase : INT32_MIN;
			if ((float)cycle_adj > phase) cycle_adj = (int32_t)((uint32_t)cycle_adj - 1u);
		} else
			cycle_adj = floorf(phase);
		uint32_t cycle = cycle_buf[i] + cycle_adj;
		phase -= cycle_adj;
		float a, b;
		ras_ends(&c, cycle, &a, &b);
		if (flags & SAU_RAS_O_PERLIN) {
			a *= perlin_amp * phase;
			b *= perlin_amp * (phase - 1.f);
		}
		if (flags & SAU_RAS_O_HALFSHAPE) {
			/* sau_maxf(a, b), sau_minf(a, b) as the build has them: maxss a,b / minss b,a --
			 * an unordered pair (NaN) comes out swapped, equal values too (signed zeros) */
			float mx = a > b ? a : b;
			float mn = b < a ? b : a;
			a = mx; b = mn;
		}
		if (flags & SAU_RAS_O_ZIGZAG) {
			float t = a; a = b; b = t;
		}
		if (flags & SAU_RAS_O_SQUARE) {
			a *= fabsf(a);
			b *= fabsf(b);
		}
		float s = shape_val(line, phase, a, b);
		main_buf[i] = s;
		/* rasg.h:277 writes (fb_s + s + prev_s) * 0.5f; every one of the six compiled
		 * sauRasG_map_*_s loops of the reference build (-ffast-math) adds the two carried
		 * values first -- one ulp apart at times, which the hashed line shapes
		 * (uwh, ncl, nhl) turn into a different sample */
		if (g_fm_forms) n->fb_s = ((n->fb_s + n->prev_s) + s) * 0.5f;
		else n->fb_s = (n->fb_s + s + n->prev_s) * 0.5f;
		n->prev_s = s;
	}
}

/* ---------------------------------------------------------------------- */
/* block evaluation (generator.c:384-729)                                  */
/* ---------------------------------------------------------------------- */

/* generator.c:384-440 */
static void combine(float *dst, size_t len, bool wave_env, bool layer,
		const float *in, const float *amp) {
	if (wave_env) {
		for (size_t i = 0; i < len; ++i) {
			float s = in[i];
			float s_amp = amp[i] * 0.5f;
			s = (s * s_amp) + fabsf(s_amp);
			if (layer) dst[i] *= s; else dst[i] = s;
		}
	} else {
		for (size_t i = 0; i < len; ++i) {
			if (layer) dst[i] += in[i] * amp[i]; else dst[i] = in[i] * amp[i];
		}
	}
}

static uint32_t eval_op(OraGen *o, Buf *bufs, uint32_t buf_len, Op *n,
		float *parent_freq, bool wave_env, bool layer);

static void eval_list(OraGen *o, Buf *bufs, uint32_t len,
		const sauProgramIDArr *ids, float *freq, bool wave_env, int layer_mode) {
	/* layer_mode: 0 = first sets, rest layer; 1 = all layer */
	for (uint32_t i = 0; i < ids->count; ++i)
		eval_op(o, bufs, len, &o->ops[ids->ids[i]], freq, wave_env,
				layer_mode ? true : (i > 0));
}

/* generator.c:448-477 */
static void eval_param(OraGen *o, Buf *bufs, uint32_t len, Ramp *par, Ramp *r_par,
		const sauProgramIDArr *mods, const sauProgramIDArr *r_mods,
		float *mulbuf, float *reused_freq, bool is_freq) {
	float *par_buf = bufs[0];
	float *freq = reused_freq ? reused_freq : (is_freq ? par_buf : NULL);
	ramp_run(par, par_buf, len, mulbuf);
	if (r_mods->count > 0) {
		float *r_buf = bufs[1];
		ramp_run(r_par, r_buf, len, mulbuf);
		eval_list(o, bufs + 2, len, r_mods, freq, true, 0);
		float *mod = bufs[2];
		for (uint32_t i = 0; i < len; ++i)
			par_buf[i] += (r_buf[i] - par_buf[i]) * mod[i];
	} else {
		ramp_skip(r_par, len);
	}
	if (mods->count > 0)
		eval_list(o, bufs, len, mods, freq, false, 1);
}

/* generator.c:479-498 */
static bool eval_selfmod_param(OraGen *o, Buf *bufs, uint32_t len, Op *n, float *freq) {
	bool filled = false;
	if (n->pma.v0 != 0.f || (n->pma.flags & SAU_LINEP_GOAL)) {
		ramp_run(&n->pma, bufs[0], len, NULL);
		filled = true;
	} else {
		ramp_skip(&n->pma, len);
	}
	for (uint32_t i = 0; i < n->apmods->count; ++i) {
		eval_op(o, bufs, len, &o->ops[n->apmods->ids[i]], freq, false, filled);
		filled = true;
	}
	return filled;
}

/* generator.c:505-541: 'A' and 'N' operators */
static void eval_amp_or_noise(OraGen *o, Buf *bufs, uint32_t len, Op *n,
		bool wave_env, bool layer) {
	float *mix = bufs[0];
	eval_param(o, bufs + 1, len, &n->amp, &n->amp2, n->amods, n->ramods,
			NULL, NULL, false);
	float *amp = bufs[1];
	float *tmp = bufs[2];
	if (n->kind == SAU_POPT_N_noise) {
		noise_run(&n->noise, tmp, len);
	} else {
		for (uint32_t i = 0; i < len; ++i) tmp[i] = 1.f;
	}
	combine(mix, len, wave_env, layer, tmp, amp);
}

/* generator.c:548-602 */
static void eval_wosc(OraGen *o, Buf *bufs, uint32_t len, Op *n,
		float *parent_freq, bool wave_env, bool layer) {
	float *mix = bufs[0];
	uint32_t *phase = (uint32_t *)bufs[1];
	float *pm = NULL, *fpm = NULL;
	eval_param(o, bufs + 2, len, &n->freq, &n->freq2, n->fmods, n->rfmods,
			parent_freq, NULL, true);
	float *freq = bufs[2];
	if (n->pmods->count > 0) {
		eval_list(o, bufs + 3, len, n->pmods, freq, false, 0);
		pm = bufs[3];
	}
	if (n->fpmods->count > 0) {
		eval_list(o, bufs + 4, len, n->fpmods, freq, false, 0);
		fpm = bufs[4];
	}
	phasor_fill(n, phase, len, freq, pm, fpm);
	eval_param(o, bufs + 3, len, &n->amp, &n->amp2, n->amods, n->ramods,
			NULL, freq, false);
	float *amp = bufs[3];
	float *tmp = bufs[4];
	if (eval_selfmod_param(o, bufs + 5, len, n, freq))
		wosc_run(n, tmp, len, phase, bufs[5]);
	else
		wosc_run(n, tmp, len, phase, NULL);
	combine(mix, len, wave_env, layer, tmp, amp);
}

/* generator.c:609-664 */
static void eval_rasg(OraGen *o, Buf *bufs, uint32_t len, Op *n,
		float *parent_freq, bool wave_env, bool layer) {
	float *mix = bufs[0];
	uint32_t *cycle = (uint32_t *)bufs[1];
	float *ras = bufs[2];
	float *pm = NULL, *fpm = NULL;
	eval_param(o, bufs + 3, len, &n->freq, &n->freq2, n->fmods, n->rfmods,
			parent_freq, NULL, true);
	float *freq = bufs[3];
	if (n->pmods->count > 0) {
		eval_list(o, bufs + 4, len, n->pmods, freq, false, 0);
		pm = bufs[4];
	}
	if (n->fpmods->count > 0) {
		eval_list(o, bufs + 5, len, n->fpmods, freq, false, 0);
		fpm = bufs[5];
	}
	cyclor_fill(n, cycle, ras, len, freq, pm, fpm);
	eval_param(o, bufs + 4, len, &n->amp, &n->amp2, n->amods, n->ramods,
			NULL, freq, false);
	float *amp = bufs[4];
	if (eval_selfmod_param(o, bufs + 5, len, n, freq))
		rasg_run_selfmod(n, len, ras, cycle, bufs[5]);
	else
		rasg_run(n, len, ras, bufs[5], bufs[6], cycle);
	combine(mix, len, wave_env, layer, ras, amp);
}

/* generator.c:675-729 */
static uint32_t eval_op(OraGen *o, Buf *bufs, uint32_t buf_len, Op *n,
		float *parent_freq, bool wave_env, bool layer) {
	float *mix = bufs[0];
	if (n->flags & OPF_VISITED) {
		for (uint32_t i = 0; i < buf_len; ++i) mix[i] = 0;
		return buf_len;
	}
	n->flags |= OPF_VISITED;
	uint32_t len = buf_len, skip_len = 0;
	if (n->time < len && !(n->flags & OPF_TIME_INF)) {
		skip_len = len - n->time;
		len = n->time;
	}
	switch (n->kind) {
	case SAU_POPT_N_amp:
	case SAU_POPT_N_noise:
		eval_amp_or_noise(o, bufs, len, n, wave_env, layer);
		break;
	case SAU_POPT_N_wave:
		eval_wosc(o, bufs, len, n, parent_freq, wave_env, layer);
		break;
	case SAU_POPT_N_raseg:
		eval_rasg(o, bufs, len, n, parent_freq, wave_env, layer);
		break;
	}
	if (!(n->flags & OPF_TIME_INF)) {
		if (!layer && skip_len > 0) {
			for (uint32_t i = 0; i < skip_len; ++i) mix[len + i] = 0;
		}
		n->time -= len;
	}
	n->flags &= ~OPF_VISITED;
	return len;
}

/* ---------------------------------------------------------------------- */
/* voices, mixdown, control loop (generator.c:734-973)                     */
/* ---------------------------------------------------------------------- */

/* generator.c:749-788 */
static void voice_mix(OraGen *o, Op *n, Voice *vn, uint32_t len) {
	float *s_buf = o->bufs[0];
	float *pan_buf = NULL;
	float *mix_l = o->mix[0], *mix_r = o->mix[1];
	if ((n->pan.flags & SAU_LINEP_GOAL) || n->camods->count > 0) {
		pan_buf = o->bufs[1 + vn->freq_buf_id];
		ramp_run(&n->pan, pan_buf, len, NULL);
	} else {
		ramp_skip(&n->pan, len);
	}
	if (n->camods->count > 0) {
		float *freq_buf = vn->freq_buf_id > 0 ? o->bufs[vn->freq_buf_id] : NULL;
		eval_list(o, o->bufs + 1 + vn->freq_buf_id, len, n->camods, freq_buf, false, 1);
	}
	for (uint32_t i = 0; i < len; ++i) {
		float s = s_buf[i] * o->amp_scale;
		float s_r = s * (pan_buf ? pan_buf[i] : n->pan.v0);
		if (g_fm_forms) { /* (L + s) - s_r, (R + s) + s_r in the reference build */
			mix_l[i] = (mix_l[i] + s) - s_r;
			mix_r[i] = (mix_r[i] + s) + s_r;
		} else {
			mix_l[i] += s - s_r;
			mix_r[i] += s + s_r;
		}
	}
	if (o->mix_used < len) o->mix_used = len;
}

/* generator.c:833-846 */
static uint32_t voice_run(OraGen *o, Voice *vn, uint32_t len) {
	Op *n = &o->ops[vn->carr_op_id];
	uint32_t time = vn->duration, out_len = 0;
	if (len > o->block_len) len = o->block_len;
	if (time > len) time = len;
	if (n->time > 0)
		out_len = eval_op(o, o->bufs, time, n, NULL, false, false);
	if (out_len > 0)
		voice_mix(o, n, vn, out_len);
	vn->duration -= time;
	return out_len;
}

static inline float clampf(float x, float lo, float hi) {
	/* sau/math.h:133-137 as the reference build has it: -ffast-math turns the two selections into
	 * minss(maxss(x, lo), hi), and maxss returns its second operand when the first is a NaN -- a NaN in
	 * the mix leaves as lo (-32767 in the PCM), in loop bodies and tails alike (pinned against
	 * oracle/_ref on programs whose feedback runs to infinity: tests/test_oracle.py) */
	if (x != x) return lo;
	x = x < lo ? lo : x;
	x = x > hi ? hi : x;
	return x;
}

/* generator.c:854-878 with 734-740 and 795-825 */
static uint32_t run_span(OraGen *o, uint32_t time, int16_t *buf, bool stereo) {
	int16_t *sp = buf;
	uint32_t gen_len = 0;
	while (time > 0) {
		uint32_t len = (time < o->block_len) ? time : o->block_len;
		time -= len;
		if (o->mix_used) {
			memset(o->mix[0], 0, sizeof(float) * o->mix_used);
			memset(o->mix[1], 0, sizeof(float) * o->mix_used);
			o->mix_used = 0;
		}
		uint32_t last_len = 0;
		for (uint32_t i = o->voice; i < o->vo_count; ++i) {
			Voice *vn = &o->voices[i];
			if (vn->duration != 0) {
				uint32_t vlen = voice_run(o, vn, len);
				if (vlen > last_len) last_len = vlen;
			}
		}
		if (last_len > 0) {
			gen_len += last_len;
			o->out_dirty_cleared = false;
			const float *l = o->mix[0], *r = o->mix[1];
			if (stereo) {
				for (uint32_t i = 0; i < last_len; ++i) {
					float sl = clampf(l[i], -1.f, 1.f);
					float sr = clampf(r[i], -1.f, 1.f);
					*sp++ += lrintf(sl * (float)INT16_MAX);
					*sp++ += lrintf(sr * (float)INT16_MAX);
				}
			} else {
				for (uint32_t i = 0; i < last_len; ++i) {
					float sm = (l[i] + r[i]) * 0.5f;
					sm = clampf(sm, -1.f, 1.f);
					*sp++ += lrintf(sm * (float)INT16_MAX);
				}
			}
		}
	}
	return gen_len;
}

/* generator.c:905-973 */
ORA_API bool ora_run(OraGen *o, int16_t *buf, size_t buf_len, bool stereo,
		size_t *out_len) {
	int16_t *sp = buf;
	uint32_t len = buf_len;
	uint32_t skip_len, last_len, gen_len = 0;
	if (!o->out_dirty_cleared) {
		o->out_dirty_cleared = true;
		memset(buf, 0, sizeof(int16_t) * (stereo ? len * 2 : len));
	}
	for (;;) {
		skip_len = 0;
		while (o->event < o->ev_count) {
			Event *e = &o->events[o->event];
			if (o->event_pos < e->wait) {
				uint32_t waittime = e->wait - o->event_pos;
				if (waittime < len) {
					skip_len = len - waittime;
					len = waittime;
				}
				o->event_pos += len;
				break;
			}
			apply_event(o, e);
			++o->event;
			o->event_pos = 0;
		}
		last_len = run_span(o, len, sp, stereo);
		if (skip_len > 0) {
			gen_len += len;
			sp += stereo ? len * 2 : len;
			len = skip_len;
			continue;
		}
		gen_len += last_len;
		break;
	}
	for (;;) {
		if (o->voice == o->vo_count) {
			if (o->event != o->ev_count) break;
			if (out_len) *out_len = gen_len;
			return false;
		}
		if (o->voices[o->voice].duration != 0) break;
		++o->voice;
	}
	if (out_len) *out_len = buf_len;
	return true;
}

/* ---- kernel-level entry points for known-answer tests ------------------- */

/** Run a free-standing W oscillator: phases from freq/pm/fpm, then output.
 * state[0]=phase accumulator, returned updated. pm_a may be NULL. */
ORA_API void ora_wosc_kat(int wave, uint32_t srate, uint32_t phase0, size_t len,
		const float *freq, const float *pm, const float *fpm, const float *pm_a,
		uint32_t *phase_out, float *out, size_t chunk) {
	ensure_tables();
	Op n;
	memset(&n, 0, sizeof n);
	n.coeff = (float)(0x1p32 / srate);
	n.wave = wave;
	n.phase = phase0 + g_picoeffs[wave].phase_adj;
	n.osc_flags = OSC_RESET;
	if (!chunk) chunk = len;
	for (size_t at = 0; at < len; at += chunk) {
		size_t m = len - at < chunk ? len - at : chunk;
		phasor_fill(&n, phase_out + at, m, freq + at, pm ? pm + at : NULL, fpm ? fpm + at : NULL);
		wosc_run(&n, out + at, m, phase_out + at, pm_a ? pm_a + at : NULL);
	}
}

ORA_API void ora_noise_kat(int type, uint32_t seed, float *out, size_t len) {
	Noise z = {seed, 0, (uint8_t)type};
	noise_run(&z, out, len);
}

/** Free-standing R oscillator; opt fields given separately. */
ORA_API void ora_rasg_kat(uint32_t srate, int line, unsigned func, unsigned flags,
		unsigned level, uint32_t alpha, uint32_t seed, uint32_t phase0,
		size_t len, const float *freq, const float *pm, const float *pm_a, float *out) {
	Op n;
	memset(&n, 0, sizeof n);
	n.coeff = (float)(0x1p32 / srate);
	n.ras.rate2x = true;
	n.ras.line = SAU_LINE_N_lin; n.ras.func = SAU_RAS_F_URAND;
	n.ras.level = ras_level(9); n.ras.alpha = 0x9e3779b9u;
	sauRasOpt opt;
	memset(&opt, 0, sizeof opt);
	opt.line = line; opt.func = func; opt.level = level; opt.alpha = alpha;
	opt.flags = flags | SAU_RAS_O_LINE_SET | SAU_RAS_O_FUNC_SET |
		SAU_RAS_O_LEVEL_SET | SAU_RAS_O_ASUBVAL_SET;
	ras_set_opt(&n.ras, &opt);
	ras_set_phase(&n.ras, phase0);
	ras_set_cycle(&n.ras, seed);
	uint32_t *cycle = malloc(len * sizeof(uint32_t));
	float *ea = malloc(len * sizeof(float)), *eb = malloc(len * sizeof(float));
	cyclor_fill(&n, cycle, out, len, freq, pm, NULL);
	if (pm_a) rasg_run_selfmod(&n, len, out, cycle, pm_a);
	else rasg_run(&n, len, out, ea, eb, cycle);
	free(cycle); free(ea); free(eb);
}
