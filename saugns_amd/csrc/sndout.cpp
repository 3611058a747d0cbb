/* sndout.cpp -- output stage: render a program straight into a sound file.
 *
 * Row f-2 of SURVEY.md section 8: the reference writes finished PCM with
 * player/sndfile.c (AU header 63-72 + size update 74-80, WAV header 82-99 +
 * size update 101-109, in-place byte swap for AU 160-168, fwrite 179-187) from
 * Player_run's synchronous chunk loop (saugns.c:589-618). Here the generator
 * produces the file's byte order on the device, and finished chunks travel to
 * page-locked memory and to the file while the next chunk renders.
 * Files are byte-identical to the reference writer's for the same PCM,
 * including its AU quirk of storing the frame count in the size field.
 */
#include "../../include/saugns_amd.h"
#include "engine.h"
#include "hip_backend.h"
#include "capi_internal.h"
#include <stdio.h>
#include <string.h>
#include <string>

using sauengine::Backend;
using sauengine::Engine;

namespace {

void put_le16(FILE *f, uint16_t v) { putc(v & 0xff, f); putc((v >> 8) & 0xff, f); }
void put_le32(FILE *f, uint32_t v) { put_le16(f, (uint16_t)(v & 0xffff)); put_le16(f, (uint16_t)(v >> 16)); }
void put_be32(FILE *f, uint32_t v) {
	putc((v >> 24) & 0xff, f); putc((v >> 16) & 0xff, f); putc((v >> 8) & 0xff, f); putc(v & 0xff, f);
}

struct SndOut {
	FILE *f = nullptr;
	int format = SAU_AMD_SNDFILE_RAW;
	uint16_t channels = 1;
	uint64_t frames = 0;

	bool open(const char *path, int fmt, uint16_t ch, uint32_t srate) {
		f = fopen(path, "wb");
		if (!f) return false;
		format = fmt; channels = ch; frames = 0;
		if (fmt == SAU_AMD_SNDFILE_AU) { /* player/sndfile.c:63-72 */
			fputs(".snd", f);
			put_be32(f, 28);
			put_be32(f, 0xffffffffu); /* size: unspecified until closed */
			put_be32(f, 3);           /* 16-bit linear PCM */
			put_be32(f, srate);
			put_be32(f, ch);
			put_be32(f, 0);
		} else if (fmt == SAU_AMD_SNDFILE_WAV) { /* player/sndfile.c:82-99 */
			fputs("RIFF", f);
			put_le32(f, 36);
			fputs("WAVE", f);
			fputs("fmt ", f);
			put_le32(f, 16);
			put_le16(f, 1);
			put_le16(f, ch);
			put_le32(f, srate);
			put_le32(f, (uint32_t)ch * srate * 2);
			put_le16(f, (uint16_t)(ch * 2));
			put_le16(f, 16);
			fputs("data", f);
			put_le32(f, 0);
		}
		return true;
	}
	bool write(const int16_t *buf, size_t n_frames) {
		size_t w = fwrite(buf, (size_t)channels * 2, n_frames, f);
		frames += w;
		return w == n_frames;
	}
	int close() {
		if (!f) return 0;
		if (format == SAU_AMD_SNDFILE_AU) { /* player/sndfile.c:74-80 */
			if (frames < UINT32_MAX) { fseek(f, 8, SEEK_SET); put_be32(f, (uint32_t)frames); }
		} else if (format == SAU_AMD_SNDFILE_WAV) { /* player/sndfile.c:101-109 */
			uint32_t bytes = (uint32_t)(channels * frames * 2);
			fseek(f, 4, SEEK_SET);
			put_le32(f, 36 + bytes);
			fseek(f, 32, SEEK_CUR);
			put_le32(f, bytes);
		}
		int err = ferror(f);
		fclose(f);
		f = nullptr;
		return err;
	}
};

thread_local std::string g_file_error;

bool render_file_over(const sauProgram *prg, uint32_t srate, const char *path, int format,
		int channels, Backend *backend /* owned */, uint64_t *frames_out, std::string &err) {
	if (!prg || !path || (channels != 1 && channels != 2) || format < 0 || format > SAU_AMD_SNDFILE_WAV) {
		err = "bad argument";
		delete backend;
		return false;
	}
	Engine *engine = Engine::create(&prg, 1, srate, backend, err);
	if (!engine) return false;
	const bool stereo = channels == 2;
	engine->set_pcm_byteswap(format == SAU_AMD_SNDFILE_AU);
	/* Player_run asks the generator for 256 ms at a time (saugns.c:471,526: ch_len); the
	 * reference's block lattice restarts at each of those calls, so a device run covers whole ones */
	size_t call = (size_t)((uint64_t)256 * srate / 1000);
	if (call == 0) call = 1;
	engine->set_call_len(call);
	const size_t chunk = call >= 176400 ? call : 176400 / call * call; /* frames per device run */
	const size_t bytes = chunk * (size_t)channels * sizeof(int16_t);
	int16_t *host[2] = {(int16_t *)backend->alloc_host(bytes), (int16_t *)backend->alloc_host(bytes)};
	SndOut out;
	bool ok = host[0] && host[1];
	if (!ok) err = "out of page-locked memory";
	if (ok && !out.open(path, format, (uint16_t)channels, srate)) {
		err = std::string("couldn't open \"") + path + "\" for writing";
		ok = false;
	}
	size_t pending[2] = {0, 0};
	int slot = 0;
	bool more = ok;
	while (ok && more) {
		size_t len = 0;
		/* PCM stays on the device; the copy below queues behind the mixer */
		ok = engine->run(nullptr, chunk, stereo, &more, &len, err);
		if (!ok) break;
		if (len) {
			ok = backend->fetch_pcm_async(0, host[slot], (uint32_t)len, stereo, slot, err);
			pending[slot] = len;
		}
		/* while this chunk renders and copies, the previous one goes to the file */
		const int other = slot ^ 1;
		if (ok && pending[other]) {
			ok = backend->wait_fetch(other, err);
			if (ok && !out.write(host[other], pending[other])) { err = "write failed"; ok = false; }
			pending[other] = 0;
		}
		slot = other;
	}
	for (int s = 0; ok && s < 2; ++s) { /* oldest first: `slot` is the older of the two */
		const int k = slot ^ s;
		if (pending[k]) {
			ok = backend->wait_fetch(k, err);
			if (ok && !out.write(host[k], pending[k])) { err = "write failed"; ok = false; }
			pending[k] = 0;
		}
	}
	(void)backend->sync(err);
	if (out.f && out.close() != 0 && ok) { err = "write failed"; ok = false; }
	if (frames_out) *frames_out = out.frames;
	backend->free_host(host[0]);
	backend->free_host(host[1]);
	delete engine; /* owns the backend */
	return ok;
}

} /* namespace */

extern "C" bool sauAmd_render_file(const sauProgram *prg, uint32_t srate, const char *path,
		int format, int channels, uint64_t *frames_out) {
	std::string err;
	sauhip::HipBackend *hip = sauhip::create_hip_backend(err);
	bool ok = hip && render_file_over(prg, srate, path, format, channels, hip, frames_out, err);
	if (!ok) {
		g_file_error = err;
		fprintf(stderr, "error [output]: %s\n", err.c_str());
	}
	return ok;
}

/* (tests/hooks: the same output stage over a caller-supplied backend -- CPU tests of headers, chunking, byte order) */
bool sauamd_internal::render_file(const sauProgram *prg, uint32_t srate, const char *path, int format, int channels,
		Backend *injected, uint64_t *frames_out, std::string &err) {
	return injected && render_file_over(prg, srate, path, format, channels, injected, frames_out, err);
}
