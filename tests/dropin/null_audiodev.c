/* null_audiodev.c -- TEST INFRASTRUCTURE: a system-audio "device" for the unchanged reference host.
 *
 * The reference's CLI (saugns.c) talks to the sound card through the four functions of
 * player/audiodev.h:23-29; its own implementations (player/audiodev/{linux,oss,sndio}.c) need ALSA /
 * OSS / sndio headers this image does not have. This file implements that interface -- written here,
 * not derived from the reference's device code -- as a device that plays into a file, so that the
 * host itself (saugns.c + player/sndfile.c, compiled from where they lie under /root/reference by
 * oracle/Makefile `hosts`) can be linked once with the reference's generator and once with
 * libsaugns_amd.so. It is never part of the product and is not used to produce oracle numbers
 * (those come from oracle/_ref/libsau_ref.so, reference sources only).
 *
 *   SAU_NULLDEV_SRATE=<Hz>  the rate the "hardware" supports (default: whatever is asked). A rate
 *                           other than the requested one makes the host run a second generator at
 *                           the device rate (saugns.c:585 split_gen) when a file is written too.
 *   SAU_NULLDEV_OUT=<path>  where the device's PCM goes (default: discarded).
 */
#include "player/audiodev.h"
#include <stdio.h>
#include <stdlib.h>

struct SGS_AudioDev {
	FILE *f;
	uint16_t channels;
	uint32_t srate;
};

SGS_AudioDev *SGS_open_AudioDev(uint16_t channels, uint32_t *restrict srate) {
	SGS_AudioDev *o = calloc(1, sizeof(SGS_AudioDev));
	if (!o) return NULL;
	const char *want = getenv("SAU_NULLDEV_SRATE");
	if (want && atol(want) > 0) *srate = (uint32_t)atol(want);
	o->channels = channels;
	o->srate = *srate;
	const char *path = getenv("SAU_NULLDEV_OUT");
	if (path) {
		o->f = fopen(path, "wb");
		if (!o->f) { free(o); return NULL; }
	}
	return o;
}

void SGS_close_AudioDev(SGS_AudioDev *restrict o) {
	if (!o) return;
	if (o->f) fclose(o->f);
	free(o);
}

uint32_t SGS_AudioDev_get_srate(const SGS_AudioDev *restrict o) { return o->srate; }

bool SGS_AudioDev_write(SGS_AudioDev *restrict o, const int16_t *restrict buf, uint32_t samples) {
	if (!o->f) return true;
	return fwrite(buf, sizeof(int16_t) * o->channels, samples, o->f) == samples;
}
