/* test_hooks.cpp -- TEST INFRASTRUCTURE: the entry points tests/ need beside the product's C ABI, in a library of their
 * own (tests/hooks/libsaugns_amd_hooks.so = the product's object files + this file + kat_kernels.hip). The product library
 * exports none of them (`nm -D saugns_amd/libsaugns_amd.so` is checked by tests/test_host.py).
 *   sauAmd_create_Generator_with_backend / sauAmd_create_Batch_with_backend / sauAmd_render_file_with_backend:
 *     the host control plane, the drop-in generator and the output stage over a caller-supplied sauengine::Backend
 *     (tests/seqexec: the sequential plan executor) -- host logic without a GPU
 *   sauAmd_Generator_rewinds: how often a changed call size / channel layout took a generator's read-ahead back
 *   sauAmd_kat_line_device / _host, sauAmd_kat_div_device: the shared arithmetic as compiled for the device and the host
 *   sauAmd_kat_scan64_device: the kernels' 64-bit wave scan and sum (k_wave_scan.h) over caller-supplied values */
#include "../../saugns_amd/csrc/capi_internal.h"
#include "../../saugns_amd/csrc/sau_dev_ops.h"
#include <stdio.h>
#include <string.h>

#define HOOK extern "C" __attribute__((visibility("default")))

#ifndef SAU_HOOKS_NO_HIP
bool kat_div(float a, int variant, unsigned long long *mismatches, uint32_t *first_bad);
bool kat_line(const saudev::LineState &st, uint32_t len, const float *mul, float *out, saudev::LineState *st_out);
bool kat_scan64(const unsigned long long *in, unsigned long long *scan, unsigned long long *sum, uint32_t n_waves);
bool kat_rint64(int scattered, unsigned long long *mismatches, uint32_t *first_bad);
#endif

HOOK sauGenerator *sauAmd_create_Generator_with_backend(const sauProgram *prg, uint32_t srate, void *backend) {
	return sauamd_internal::make_generator(prg, srate, (sauengine::Backend *)backend);
}
HOOK sauAmdBatch *sauAmd_create_Batch_with_backend(const sauProgram *const *prgs, size_t n, uint32_t srate, void *backend) {
	if (!backend) return nullptr;
	return sauamd_internal::make_batch_over(prgs, n, srate, (sauengine::Backend *)backend);
}
HOOK bool sauAmd_render_file_with_backend(const sauProgram *prg, uint32_t srate, const char *path, int format, int channels,
		void *backend, uint64_t *frames_out) {
	std::string err;
	const bool ok = sauamd_internal::render_file(prg, srate, path, format, channels, (sauengine::Backend *)backend, frames_out, err);
	if (!ok) fprintf(stderr, "error [output]: %s\n", err.c_str());
	return ok;
}
HOOK unsigned sauAmd_Generator_rewinds(const sauGenerator *g) { return sauamd_internal::generator_rewinds(g); }

/* state = {v0, vt, pos, end, type, flags} as 6 dwords, updated in place */
HOOK int sauAmd_kat_line_host(uint32_t *state, uint32_t len, const float *mul, float *out) {
	saudev::LineState st;
	memcpy(&st, state, sizeof st);
	saudev::LineBlock lb = saudev::line_begin(st, len, mul != nullptr, mul ? mul[0] : 0.f, saudev::lattice_none(), 0);
	for (uint32_t j = 0; j < len; ++j) out[j] = saudev::line_value(lb, j, mul ? mul[j] : 1.f);
	memcpy(state, &st, sizeof st);
	return 1;
}
#ifndef SAU_HOOKS_NO_HIP
HOOK int sauAmd_kat_line_device(uint32_t *state, uint32_t len, const float *mul, float *out) {
	saudev::LineState st, st2;
	memcpy(&st, state, sizeof st);
	if (!kat_line(st, len, mul, out, &st2)) return 0;
	memcpy(state, &st2, sizeof st2);
	return 1;
}
/* for wave id w: how many divisors make div_diff_scale differ from IEEE division (and the first such bit pattern);
 * variant 1 is the uncorrected product a * rcp(b), which the probe must catch. -1 on a device error. */
HOOK long long sauAmd_kat_div_device(uint32_t wave, int variant, uint32_t *first_bad) {
	unsigned long long m = 0;
	uint32_t fb = 0xffffffffu;
	if (wave >= 12 || !kat_div(sauengine::wave_consts()[wave].diff_scale, variant, &m, &fb)) return -1;
	if (first_bad) *first_bad = fb;
	return (long long)m;
}
/* in[64 * n_waves] -> scan[64 * n_waves] (inclusive per wave), sum[n_waves]; 0 on a device error */
/* rint64() and franssgauss32()'s conversions over every 32-bit pattern (kat_kernels.hip): mismatches, or -1 */
HOOK long long sauAmd_kat_rint64_device(int scattered, uint32_t *first_bad) {
	unsigned long long m = 0;
	uint32_t fb = 0;
	if (!kat_rint64(scattered, &m, &fb)) return -1;
	if (first_bad) *first_bad = fb;
	return (long long)m;
}
HOOK int sauAmd_kat_scan64_device(const unsigned long long *in, unsigned long long *scan, unsigned long long *sum, uint32_t n_waves) {
	return kat_scan64(in, scan, sum, n_waves) ? 1 : 0;
}
#endif
