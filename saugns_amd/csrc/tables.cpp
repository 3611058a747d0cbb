/* tables.cpp -- the twelve pre-integrated wave tables ("PILUTs") and their
 * per-wave constants.
 *
 * Table *generation* is host-side, one-time work (sau/wave.c:77-221); only the
 * lookup is on the hot path.  Three sources, in order of preference:
 *   1. tables handed in through sauAmd_set_piluts() (tests use the compiled
 *      reference's own arrays, which differ from a strict-order build by 1 ulp
 *      in srs/ean/cat/mto because the reference compiles wave.c with
 *      -ffast-math),
 *   2. when this library is linked into the reference host: that host's own
 *      `sauWave_piluts` (looked up with dlsym, so the generator uses exactly
 *      the tables the rest of the program was built with),
 *   3. the strict-order builder below.
 */
#include "engine.h"
#include <dlfcn.h>
#include <math.h>
#include <string.h>

namespace sauengine {

namespace {

enum { WLEN = 2048, WHALF = WLEN / 2, WQUART = WLEN / 4 };

float g_tables[SAU_WAVE_NAMED][WLEN];
bool g_ready = false;

/* sau/wave.h:33-69: {amp_scale, amp_dc, phase_adj} */
struct PiCoeff { float amp_scale, amp_dc; int32_t phase_adj; };
const PiCoeff g_pico[SAU_WAVE_NAMED] = {
	{1.27324153848f, 0.0f, INT32_MIN / 2},            /* sin */
	{1.00097751711f, 0.0f, 0},                        /* tri */
	{1.52547437578f, 0.0f, 0},                        /* srs */
	{2.00000000000f, 0.0f, INT32_MIN / 2},            /* sqr */
	{1.20275515347f, -0.24257955076f, 0},             /* ean */
	{1.37070880305f, -0.23725526633f, 0},             /* cat */
	{(float)(1.26113986272 * -1), 0.0f, -(INT32_MIN / 2)}, /* eto */
	{1.02639326795f, -0.33333333333f, 0},             /* par */
	{1.57268451738f, -0.23724704918f, 0},             /* mto */
	{(float)(1.00048851979 * -1), 0.0f, -(INT32_MIN / 2)}, /* saw */
	{1.40333871035f, -0.36334126990f, 0},             /* hsi */
	{1.07213756312f, 0.27322393756f, 0},              /* spa */
};
WaveConst g_wconst[SAU_WAVE_NAMED];
bool g_wconst_ready = false;

/* sau/wave.c:77-98 */
void integrate(float *dst, const float *src) {
	const float inv = 1.f / (WLEN * 0.125f);
	double mean = 0.f;
	for (int i = 0; i < WLEN; ++i) mean += src[i];
	mean /= WLEN;
	double run = 0.f;
	float lo = 0.f, hi = 0.f;
	for (int i = 0; i < WLEN; ++i) {
		run += src[i] - mean;
		float x = (float)(run * inv);
		if (x < lo) lo = x;
		if (x > hi) hi = x;
		dst[i] = x;
	}
	float gain = 1.f / ((hi - lo) * 0.5f);
	float shift = -(hi + lo) * 0.5f;
	for (int i = 0; i < WLEN; ++i) dst[i] = (dst[i] + shift) * gain;
}

/* sau/wave.c:105-214, only what the PILUT set needs */
void build() {
	static float sine[WLEN], tri[WLEN], tri_i[WLEN], ean[WLEN], par[WLEN];
	static float srs[WLEN], cat[WLEN], mto[WLEN], hsi[WLEN], spa[WLEN];
	const double pi = 3.14159265358979323846;
	for (int i = 0; i < WHALF; ++i) {
		const double x = i * (1.f / WHALF);
		const float sx = (float)sin(pi * x);
		sine[i] = sx; sine[i + WHALF] = -sx;
		const float rx = sqrtf(sx);
		srs[i] = rx;
		hsi[i] = sx * 2 - 1.f;
		mto[i] = rx * 2 - 1.f;
		const float px = (float)sin(pi * 0.5f * (1 + x));
		spa[i + WQUART] = px * 2 - 1.f;
		const double xr = (WHALF - i) * (1.f / WHALF);
		par[i + WQUART] = (float)((xr * xr) * 2.f - 1.f);
	}
	par[WHALF + WQUART] = -1.f;
	spa[WHALF + WQUART] = -1.f;
	for (int i = 0; i < WQUART; ++i) {
		const double x = i * (1.f / WQUART);
		const double xr = (WQUART - i) * (1.f / WQUART);
		tri_i[i] = (float)((x * x) - 1.f);
		tri_i[i + WQUART] = (float)(1.f - (xr * xr));
		tri[i] = (float)x;
		tri[i + WQUART] = (float)xr;
		par[i] = par[WHALF - i];
		par[i + WHALF + WQUART] = par[WHALF + WQUART - i];
		spa[i] = spa[WHALF - i];
		spa[i + WHALF + WQUART] = spa[WHALF + WQUART - i];
	}
	for (int i = WHALF; i < WLEN; ++i) {
		tri_i[i] = -tri_i[i - WHALF];
		tri[i] = -tri[i - WHALF];
		hsi[i] = -1.f;
		mto[i] = -1.f;
		srs[i] = -srs[i - WHALF];
	}
	const float ean_dc = (float)((1.14603185654 - 1.f) / 2.f);
	const float ean_gain = (float)(1.f / 1.07301592827);
	for (int i = 0; i < WLEN; ++i) {
		ean[i] = (sine[i] + par[i] - tri[i] + ean_dc) * ean_gain;
		cat[i] = sine[i] + mto[i] - srs[i];
	}
	/* sau/wave.c:49-62: which array differentiates into which wave */
	memcpy(g_tables[SAU_WAVE_N_sin], sine, sizeof sine);
	memcpy(g_tables[SAU_WAVE_N_tri], tri_i, sizeof tri_i);
	integrate(g_tables[SAU_WAVE_N_srs], srs);
	memcpy(g_tables[SAU_WAVE_N_sqr], tri, sizeof tri);
	integrate(g_tables[SAU_WAVE_N_ean], ean);
	integrate(g_tables[SAU_WAVE_N_cat], cat);
	memcpy(g_tables[SAU_WAVE_N_eto], ean, sizeof ean);
	integrate(g_tables[SAU_WAVE_N_par], par);
	integrate(g_tables[SAU_WAVE_N_mto], mto);
	memcpy(g_tables[SAU_WAVE_N_saw], par, sizeof par);
	integrate(g_tables[SAU_WAVE_N_hsi], hsi);
	integrate(g_tables[SAU_WAVE_N_spa], spa);
}

/* When the reference host is in the process image, adopt its tables. */
bool adopt_host_tables() {
	typedef void (*init_f)(void);
	init_f init = (init_f)dlsym(RTLD_DEFAULT, "sau_global_init_Wave");
	float *const *tabs = (float *const *)dlsym(RTLD_DEFAULT, "sauWave_piluts");
	if (!init || !tabs) return false;
	init();
	for (int w = 0; w < SAU_WAVE_NAMED; ++w) {
		if (!tabs[w]) return false;
		memcpy(g_tables[w], tabs[w], sizeof g_tables[w]);
	}
	return true;
}

} /* namespace */

const float *builtin_piluts() {
	if (!g_ready) {
		if (!adopt_host_tables()) build();
		g_ready = true;
	}
	return &g_tables[0][0];
}

void override_piluts(const float *tables) {
	memcpy(g_tables, tables, sizeof g_tables);
	g_ready = true;
}

const WaveConst *wave_consts() {
	if (!g_wconst_ready) {
		for (int w = 0; w < SAU_WAVE_NAMED; ++w) {
			/* sau/wave.h:144-149 */
			g_wconst[w].diff_scale = g_pico[w].amp_scale * 0.125f * (float)UINT32_MAX;
			g_wconst[w].diff_offset = g_pico[w].amp_dc;
			g_wconst[w].phase_adj = g_pico[w].phase_adj;
			g_wconst[w].pad = 0;
		}
		g_wconst_ready = true;
	}
	return g_wconst;
}

} /* namespace sauengine */
