/* seq_backend.cpp -- TEST INFRASTRUCTURE: a sequential executor of the plan
 * format, plugged into the host control plane through
 * sauAmd_create_Batch_with_backend().  It lets `pytest -m "not gpu"` check
 * event handling, plan compilation and the shared per-sample arithmetic
 * (saugns_amd/csrc/sau_dev_math.h) against the oracle without a GPU.  It is
 * not part of the product and is never loaded by saugns_amd/.
 */
#include "../../saugns_amd/csrc/engine.h"
#include "../../saugns_amd/csrc/sau_dev_ops.h"
#include <string.h>
#include <vector>

using namespace saudev;
using namespace sauengine;

namespace {

/* what the host control plane asked of the most recent segment (host-logic tests) */
static uint32_t g_last_counts[5];

struct SeqBackend : public Backend {
	BackendConfig cfg;
	uint32_t block;
	std::vector<DevOp> ops;
	std::vector<Step> steps;
	std::vector<uint32_t> op_ids;
	std::vector<HerpC23> c23;
	std::vector<HerpC01> c01;
	WaveConst wc[12];
	std::vector<std::vector<int16_t>> pcm;
	uint32_t max_frames = 0;
	/* a stand-in for a device's memory (tests/test_host.py: two "devices" with budgets of their own): what was free when it was
	 * opened (0: not told), failed allocations of chain rows, segments rendered and the longest of them */
	size_t free_hint = 0;
	unsigned row_failures = 0;
	uint32_t segments = 0, longest = 0;
	size_t chain_rows_budget() override { return sauengine::chain_rows_budget(free_hint, row_failures); }

	explicit SeqBackend(uint32_t b) : block(b) {}

	bool init(const BackendConfig &c, std::string &) override {
		cfg = c;
		ops.assign(c.op_count ? c.op_count : 1, DevOp());
		memset((void *)ops.data(), 0, ops.size() * sizeof(DevOp)); /* as the device buffer starts: all zero bytes */
		c23.resize(12 * WAVE_LEN); c01.resize(12 * WAVE_LEN);
		for (uint32_t w = 0; w < 12; ++w)
			for (uint32_t i = 0; i < WAVE_LEN; ++i)
				herp_coeffs(c.piluts + (size_t)w * WAVE_LEN, i, c23[w * WAVE_LEN + i], c01[w * WAVE_LEN + i]);
		memcpy(wc, c.wconst, sizeof wc);
		pcm.resize(c.n_streams);
		return true;
	}
	bool reserve_frames(uint32_t n, bool, std::string &) override {
		max_frames = n;
		/* (poisoned, not zero: what a run leaves unwritten -- neither mixed nor cleared by Engine::render_segment -- shows) */
		for (auto &p : pcm) p.assign((size_t)n * 2, (int16_t)0x5a5a);
		return true;
	}
	bool upload_plans(const Step *s, const FastIds *, size_t ns, const uint32_t *ids, size_t ni, std::string &) override {
		steps.assign(s, s + ns); op_ids.assign(ids, ids + ni);
		return true;
	}
	bool apply_updates(const OpUpdate *r, size_t n, std::string &) override {
		for (size_t i = 0; i < n; ++i) apply_update(ops[r[i].op], r[i], wc);
		return true;
	}
	bool zero_pcm(uint32_t s0, uint32_t ns, uint32_t first, uint32_t n, bool stereo, std::string &) override {
		const size_t ch = stereo ? 2 : 1;
		for (uint32_t s = s0; s < s0 + ns; ++s) std::fill(pcm[s].begin() + (size_t)first * ch, pcm[s].begin() + (size_t)(first + n) * ch, 0);
		return true;
	}
	double herp(uint32_t wave, uint32_t phase) const {
		uint32_t ind = phase >> SLEN_BITS;
		return herp_poly(c23[wave * WAVE_LEN + ind], c01[wave * WAVE_LEN + ind], phase);
	}

	/* one voice, one segment */
	void run_voice(const VoiceDesc &vd, uint32_t seg_len, std::vector<float> &vrow,
			std::vector<float> &prow, float &pan_const, uint32_t n_slots, uint32_t n_main) {
		std::vector<DevOp> lo(vd.nops);
		for (uint32_t i = 0; i < vd.nops; ++i) lo[i] = ops[op_ids[vd.ops_ofs + i]];
		std::vector<std::vector<float>> slot(n_slots, std::vector<float>(block, 0.f));
		std::vector<uint32_t> tmpu(block);
		vrow.assign(seg_len, 0.f);
		uint32_t done = 0, produced = 0;
		uint32_t stack[MAX_NEST + 2], rstack[MAX_NEST + 2];
		while (done < vd.run_len) {
			if (lo[vd.carr_local].time == 0) break;
			uint32_t blen = std::min(block, vd.run_len - done);
			uint32_t depth = 0, cur_len = blen;
			/* frames until the operator being evaluated, an ancestor or the voice stops (the reference cuts its blocks there) */
			uint32_t cur_rem = (vd.flags & VD_MORE) ? TAIL_FAR : std::min(vd.run_len - done, TAIL_FAR);
			bool ended = false;
			for (uint32_t si = 0; si < vd.plan_len && !ended; ++si) {
				/* slot ids -> memory indices (wide plans: step pairs, sau_dev_types.h) */
				const Step &st_lo = steps[vd.plan_ofs + si];
				const bool wide = (vd.flags & VD_WIDE) != 0;
				const WideStep st = step_widen(st_lo, wide ? &steps[vd.plan_ofs + si + 1] : nullptr, n_main);
				if (wide) ++si;
				DevOp &op = lo[st.op];
				uint32_t parent_len = cur_len;
				if (st.flags & SF_BEGIN) {
					rstack[depth] = cur_rem;
					stack[depth++] = cur_len;
					if (!(op.flags & OPF_TIME_INF) && op.time < cur_len) cur_len = op.time;
					if (!(op.flags & OPF_TIME_INF) && op.time < cur_rem) cur_rem = op.time;
				}
				const uint32_t len = cur_len;
				TailCtx tc;
				tc.lat = vd.lat; tc.ev_left = vd.ev_left; tc.off = done; tc.rem = cur_rem; tc.on = (vd.flags & VD_TAILS) ? 1u : 0u;
				switch (st.kind) {
				case ST_ZERO:
					for (uint32_t j = 0; j < len; ++j) slot[st.out][j] = 0.f;
					break;
				case ST_LINE: {
					const float *mul = st.fmul != NO_WSLOT ? slot[st.fmul].data() : nullptr;
					LineBlock lb = line_begin(op.line[st.which], len, mul != nullptr, mul ? mul[0] : 0.f, vd.lat, done);
					for (uint32_t j = 0; j < len; ++j)
						slot[st.out][j] = line_value_t(lb, j, mul ? mul[j] : 1.f, tc);
					if (st.flags & SF_SKIP2) line_skip(op.line[st.tmp], len, vd.lat, done);
					break;
				}
				case ST_SMLINE: {
					LineState &ls = op.line[L_PMA];
					bool active = (ls.v0 != 0.f) || (ls.flags & LP_GOAL);
					if (active) {
						LineBlock lb = line_begin(ls, len, false, 0.f, vd.lat, done);
						for (uint32_t j = 0; j < len; ++j) slot[st.out][j] = line_value_t(lb, j, 1.f, tc);
					} else {
						line_skip(ls, len, vd.lat, done);
						for (uint32_t j = 0; j < len; ++j) slot[st.out][j] = 0.f;
					}
					break;
				}
				case ST_LERP:
					for (uint32_t j = 0; j < len; ++j) {
						float pv = slot[st.out][j];
						pv += (slot[st.freq][j] - pv) * slot[st.pm][j];
						slot[st.out][j] = pv;
					}
					break;
				case ST_OSC: {
					const float *fslot = st.freq != NO_WSLOT ? slot[st.freq].data() : nullptr;
					const float *fmul = st.fmul != NO_WSLOT ? slot[st.fmul].data() : nullptr;
					const float *pmS = st.pm != NO_WSLOT ? slot[st.pm].data() : nullptr;
					const float *fpmS = st.fpm != NO_WSLOT ? slot[st.fpm].data() : nullptr;
					const float *ampS = st.amp != NO_WSLOT ? slot[st.amp].data() : nullptr;
					const float *smS = st.sm != NO_WSLOT ? slot[st.sm].data() : nullptr;
					const uint32_t type = op.type;
					const bool is_osc = type == OT_WAVE || type == OT_RASEG;
					std::vector<float> fv(len), av(len), pv(len), s(len);
					if (is_osc) {
						if (fslot) for (uint32_t j = 0; j < len; ++j) fv[j] = fslot[j];
						else {
							LineBlock lb = line_begin(op.line[L_FREQ], len, fmul != nullptr, fmul ? fmul[0] : 0.f, vd.lat, done);
							for (uint32_t j = 0; j < len; ++j) fv[j] = line_value_t(lb, j, fmul ? fmul[j] : 1.f, tc);
							line_skip(op.line[L_FREQ2], len, vd.lat, done);
						}
					}
					if (ampS) for (uint32_t j = 0; j < len; ++j) av[j] = ampS[j];
					else {
						LineBlock lb = line_begin(op.line[L_AMP], len, false, 0.f, vd.lat, done);
						for (uint32_t j = 0; j < len; ++j) av[j] = line_value_t(lb, j, 1.f, tc);
						line_skip(op.line[L_AMP2], len, vd.lat, done);
					}
					bool selfmod = is_osc && smS != nullptr;
					if (is_osc && (st.flags & SF_SM_INLINE)) {
						LineState &pl = op.line[L_PMA];
						if ((pl.v0 != 0.f) || (pl.flags & LP_GOAL)) {
							LineBlock lb = line_begin(pl, len, false, 0.f, vd.lat, done);
							for (uint32_t j = 0; j < len; ++j) pv[j] = line_value_t(lb, j, 1.f, tc);
							selfmod = true;
						} else line_skip(pl, len, vd.lat, done);
					} else if (smS) for (uint32_t j = 0; j < len; ++j) pv[j] = smS[j];
					if (type == OT_WAVE) {
						const WaveConst &k = wc[op.wave];
						for (uint32_t j = 0; j < len; ++j) { /* wosc.h:135-169 */
							uint32_t ofs = (uint32_t)pm_offset(pmS != nullptr, fpmS != nullptr,
									pmS ? pmS[j] : 0.f, fpmS ? fpmS[j] : 0.f, fv[j], 0x1p31f);
							op.phase += rint32w(op.coeff * fv[j]);
							tmpu[j] = ofs + op.phase;
						}
						if (len > 0 && (op.flags & OPF_OSC_RESET)) { /* wosc.h:215-231 */
							op.prev_Is = herp(op.wave, tmpu[0] - SLEN);
							double Is0 = herp(op.wave, tmpu[0]);
							{ /* the reference build's form of the restart sample (sau_dev_math.h: wosc_reset_s) */
								const uint32_t pp = tmpu[0] - SLEN, ip = op.wave * WAVE_LEN + (pp >> SLEN_BITS);
								op.prev_s = wosc_reset_s(Is0, herp_poly_rise(c23[ip], c01[ip], pp), c01[ip].c0, k.diff_scale, k.diff_offset);
							}
							op.prev_Is = Is0; op.prev_phase = tmpu[0];
							op.flags &= ~OPF_OSC_RESET;
						}
						for (uint32_t j = 0; j < len; ++j) { /* wosc.h:238-310 */
							uint32_t phase = tmpu[j];
							if (selfmod) phase += rint32w(op.fb_s * pv[j] * 0x1p31f);
							int32_t d = (int32_t)(phase - op.prev_phase);
							float sv;
							if (d == 0) sv = op.prev_s;
							else {
								double Is = herp(op.wave, phase);
								sv = wosc_diff(Is, op.prev_Is, d, k.diff_scale, k.diff_offset);
								op.prev_Is = Is; op.prev_s = sv; op.prev_phase = phase;
							}
							s[j] = sv;
							if (selfmod) op.fb_s = (op.fb_s + sv) * 0.5f;
						}
					} else if (type == OT_RASEG) {
						const bool r2 = (op.flags & OPF_RATE2X) != 0;
						const float coeff = r2 ? op.coeff * 2 : op.coeff;
						const float pscale = r2 ? 0x1p31f * 2 : 0x1p31f;
						RasParams rp = ras_params(op.ras_func, op.ras_flags, op.ras_level, op.ras_alpha, op.wave);
						for (uint32_t j = 0; j < len; ++j) {
							uint64_t ofs = (uint64_t)pm_offset(pmS != nullptr, fpmS != nullptr,
									pmS ? pmS[j] : 0.f, fpmS ? fpmS[j] : 0.f, fv[j], pscale);
							uint64_t cp = ofs + op.cycle_phase;
							op.cycle_phase += (uint64_t)rint64(coeff * fv[j]);
							uint32_t cyc; float ph;
							ras_split(cp, cyc, ph);
							if (!selfmod) s[j] = ras_sample(rp, cyc, ph, true, rp.line == LN_cub && cub_map_is_tail(tc, j));
							else { /* rasg.h:242-280 */
								float pm_a = ras_fb_amount(op.fb_s, pv[j]);
								float phase = ph + pm_a;
								int32_t adj = floor_i32_ref(phase);
								uint32_t cycle = cyc + (uint32_t)adj;
								phase -= (float)adj;
								float sv = ras_sample(rp, cycle, phase, false);
								s[j] = sv;
								op.fb_s = ((op.fb_s + op.prev_s) + sv) * 0.5f;
								op.prev_s = sv;
							}
						}
					} else if (type == OT_NOISE) {
						for (uint32_t j = 0; j < len; ++j) {
							uint32_t n = op.noise_n++;
							switch (op.wave) {
							case NZ_re: {
								int32_t r = (int32_t)ranfast32(n);
								op.noise_prev += (uint32_t)(r >> 6);
								s[j] = fscalei((uint32_t)foldhd32((int32_t)op.noise_prev), 0x1p-31f);
								break;
							}
							case NZ_vi: {
								uint32_t s1 = ranfast32(n);
								s[j] = fscalei((s1 / 2) - (op.noise_prev / 2), 0x1p-31f);
								op.noise_prev = s1;
								break;
							}
							case NZ_bv: {
								int32_t s1 = noise_bv_term(n);
								s[j] = (float)(s1 - (int32_t)op.noise_prev);
								op.noise_prev = (uint32_t)s1;
								break;
							}
							default: s[j] = noise_stateless(op.wave, n); break;
							}
						}
					} else {
						for (uint32_t j = 0; j < len; ++j) s[j] = 1.f;
					}
					const bool wave_env = st.flags & SF_WAVE_ENV, layer = st.flags & SF_LAYER;
					for (uint32_t j = 0; j < len; ++j)
						slot[st.out][j] = mix_combine(layer ? slot[st.out][j] : 0.f, s[j], av[j], wave_env, layer);
					if (st.which & OX_VOICE) { /* carrier hands its block to the mixer in this step */
						LineState &pl = op.line[L_PAN];
						LineBlock lb;
						bool goal = (pl.flags & LP_GOAL) != 0;
						if (goal) lb = line_begin(pl, len, false, 0.f, vd.lat, done); else line_skip(pl, len, vd.lat, done);
						for (uint32_t j = 0; j < len; ++j) {
							vrow[done + j] = slot[st.out][j];
							if (!prow.empty()) prow[done + j] = goal ? line_value_t(lb, j, 1.f, tc) : pl.v0;
						}
						produced += len;
					}
					break;
				}
				case ST_VOICE: {
					LineState &pl = op.line[L_PAN];
					const float *panS = st.pm != NO_WSLOT ? slot[st.pm].data() : nullptr;
					LineBlock lb;
					bool goal = !panS && (pl.flags & LP_GOAL);
					if (!panS) { if (goal) lb = line_begin(pl, len, false, 0.f, vd.lat, done); else line_skip(pl, len, vd.lat, done); }
					for (uint32_t j = 0; j < len; ++j) {
						vrow[done + j] = slot[st.out][j];
						if (!prow.empty())
							prow[done + j] = panS ? panS[j] : (goal ? line_value_t(lb, j, 1.f, tc) : pl.v0);
					}
					produced += len;
					break;
				}
				}
				if (st.flags & SF_END) {
					if (!(op.flags & OPF_TIME_INF)) {
						uint32_t outer = (st.flags & SF_BEGIN) ? parent_len : stack[depth - 1];
						if (!(st.flags & SF_LAYER))
							for (uint32_t j = len; j < outer; ++j) slot[st.out][j] = 0.f;
						op.time -= len;
					}
					--depth;
					/* the carrier's end sets the length of the voice-level steps (generator.c:839-846);
					 * a pan modulator ending at this level gives the length back (generator.c:762-771) */
					if (depth == 0 && st.op == vd.carr_local) { cur_len = len; if (len == 0) ended = true; }
					else { cur_len = stack[depth]; cur_rem = rstack[depth]; }
				}
			}
			done += blen;
		}
		(void)produced;
		pan_const = lo[vd.carr_local].line[L_PAN].v0;
		for (uint32_t i = 0; i < vd.nops; ++i) ops[op_ids[vd.ops_ofs + i]] = lo[i];
	}

	bool render(const SegmentDesc &seg, std::string &) override {
		++segments; if (seg.len > longest) longest = seg.len;
		g_last_counts[0] = seg.n_main; g_last_counts[1] = seg.n_fast; g_last_counts[2] = seg.n_fast_full;
		g_last_counts[3] = seg.may_scan ? 1u : 0u; g_last_counts[4] = seg.serial ? 1u : 0u;
		std::vector<std::vector<float>> vout(seg.n_voices), pan(seg.n_voices);
		std::vector<float> pan_const(seg.n_voices, 0.f);
		for (uint32_t v = 0; v < seg.n_voices; ++v) {
			if (seg.voices[v].pan_dynamic_row != ~0u) pan[v].assign(seg.len, 0.f);
			run_voice(seg.voices[v], seg.len, vout[v], pan[v], pan_const[v], seg.n_slots, seg.n_main);
		}
		for (uint32_t s = 0; s < seg.n_streams; ++s) {
			const SegmentDesc::Stream &sd = seg.streams[s];
			for (uint32_t i = 0; i < sd.write_len; ++i) {
				float L = 0.f, R = 0.f;
				for (uint32_t r = sd.first_voice; r < sd.first_voice + sd.n_voices; ++r) {
					float sv = vout[r][i] * sd.amp_scale;
					float p = pan[r].empty() ? pan_const[r] : pan[r][i];
					float s_r = sv * p;
					L = (L + sv) - s_r;
					R = (R + sv) + s_r;
				}
				if (seg.stereo) {
					int16_t l16 = pcm16(L), r16 = pcm16(R);
					pcm[s][2 * (size_t)(seg.pcm_offset + i)] = seg.swap_bytes ? pcm_swap(l16) : l16;
					pcm[s][2 * (size_t)(seg.pcm_offset + i) + 1] = seg.swap_bytes ? pcm_swap(r16) : r16;
				} else {
					int16_t m16 = pcm16((L + R) * 0.5f);
					pcm[s][seg.pcm_offset + i] = seg.swap_bytes ? pcm_swap(m16) : m16;
				}
			}
		}
		return true;
	}
	bool fetch_pcm(uint32_t s, int16_t *dst, uint32_t frames, bool stereo, std::string &) override {
		memcpy(dst, pcm[s].data(), (size_t)frames * (stereo ? 2 : 1) * sizeof(int16_t));
		return true;
	}
	const int16_t *device_pcm(uint32_t) override { return nullptr; }
	bool sync(std::string &) override { return true; }
	std::vector<DevOp> snap[4];
	bool save_state(int slot, std::string &) override { snap[slot & 3] = ops; return true; }
	bool load_state(int slot, std::string &err) override {
		if (snap[slot & 3].size() != ops.size()) { err = "no operator state was saved under this slot"; return false; }
		ops = snap[slot & 3];
		return true;
	}
};

} /* namespace */

extern "C" __attribute__((visibility("default"))) void seq_backend_last_counts(uint32_t *out5) {
	for (int i = 0; i < 5; ++i) out5[i] = g_last_counts[i];
}

extern "C" __attribute__((visibility("default"))) void *seq_backend_create(uint32_t block_len) {
	return new SeqBackend(block_len ? block_len : 1024);
}
/* the backend as a device with `free_bytes` free and `failures` failed allocations of chain rows so far */
extern "C" __attribute__((visibility("default"))) void seq_backend_set_memory(void *be, unsigned long long free_bytes, unsigned failures) {
	((SeqBackend *)be)->free_hint = (size_t)free_bytes; ((SeqBackend *)be)->row_failures = failures;
}
/* out2: segments rendered so far, the longest of them in frames (read before the batch that owns the backend is closed) */
extern "C" __attribute__((visibility("default"))) void seq_backend_segments(void *be, uint32_t *out2) {
	out2[0] = ((SeqBackend *)be)->segments; out2[1] = ((SeqBackend *)be)->longest;
}
