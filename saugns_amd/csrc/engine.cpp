/* engine.cpp -- host control plane (see engine.h). */
#include <atomic>
#include "engine.h"
#include <algorithm>
#include <exception>
#include <chrono>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

namespace sauengine {

static const sauProgramIDArr g_no_ids = {0};

/* sau/math.h:35-46 */
static uint64_t ms_to_samples(uint64_t ms, uint64_t srate, int *carry) {
	uint64_t t = ms * srate;
	if (carry) {
		t += *carry;
		*carry = (int)(t % 1000);
	}
	return t / 1000;
}

Engine::~Engine() { delete backend_; }

const char *tune_env(const char *name) {
	static const bool on = getenv("SAU_AMD_TUNE") != nullptr;
	return on ? getenv(name) : nullptr;
}

uint32_t chain_seg_frames(size_t n_chains, size_t budget) {
	if (!n_chains) return CHAIN_SEG;
	const size_t f = budget / (n_chains * 8);
	if (f <= CHAIN_SEG) return CHAIN_SEG;
	return f > (1u << 22) ? (1u << 22) : (uint32_t)f & ~63u;
}

/* generator.c:135-217 */
Engine *Engine::create(const sauProgram *const *prgs, size_t n_prgs, uint32_t srate,
		Backend *backend, std::string &err) {
	Engine *e = new Engine();
	e->backend_ = backend;
	e->srate_ = srate;
	/* (no C++ exception may reach the C ABI above: a program whose counts cannot be allocated -- a hand-made image -- is a
	 * failed constructor like any other, generator.c:200-217) */
	try {
	e->streams_.resize(n_prgs);
	uint32_t op_base = 0, vo_base = 0;
	for (size_t s = 0; s < n_prgs; ++s) {
		const sauProgram *prg = prgs[s];
		Stream &st = e->streams_[s];
		st.prg = prg;
		st.op_base = op_base;
		st.vo_base = vo_base;
		st.ops.resize(prg->op_count);
		st.voices.resize(prg->vo_count);
		st.amp_scale = 0.5f * prg->ampmult;
		if (prg->mode & SAU_PMODE_AMP_DIV_VOICES)
			st.amp_scale /= prg->vo_count;
		int carry = 0;
		st.events.resize(prg->ev_count);
		for (size_t i = 0; i < prg->ev_count; ++i) {
			st.events[i].wait = (uint32_t)ms_to_samples(prg->events[i].wait_ms, srate, &carry);
			st.events[i].pe = &prg->events[i];
		}
		op_base += prg->op_count;
		vo_base += prg->vo_count;
	}
	e->total_ops_ = op_base;
	e->total_voices_ = vo_base;
	/* the reference build's loop tails (`cub` lines: sau_dev_math.h, TailCtx); SAU_AMD_LOOP_TAILS=0: the loop bodies' forms everywhere */
	if (const char *lt = getenv("SAU_AMD_LOOP_TAILS")) e->loop_tails_ = atoi(lt) != 0;
	e->plan_cache_ = tune_env("SAU_AMD_NO_PLAN_CACHE") == nullptr;
	e->plan_check_ = tune_env("SAU_AMD_PLAN_CHECK") != nullptr; /* tests: every cached plan against a fresh compile */
	e->plan_refs_.resize(vo_base);
	BackendConfig cfg;
	cfg.srate = srate;
	cfg.op_count = op_base;
	cfg.voice_count = vo_base;
	cfg.n_streams = (uint32_t)n_prgs;
	cfg.max_frames = 0;
	cfg.piluts = builtin_piluts();
	cfg.wconst = wave_consts();
	if (!backend->init(cfg, err)) {
		delete e;
		return nullptr;
	}
	} catch (const std::exception &ex) {
		err = std::string("out of memory for the program's operator / voice / event counts (") + ex.what() + ")";
		delete e;
		return nullptr;
	}
	return e;
}

static LineUpdate make_line_update(const sauLine *src, uint32_t srate) {
	LineUpdate u;
	memset(&u, 0, sizeof u);
	if (!src) return u;
	u.v0 = src->v0; u.vt = src->vt;
	u.end_samples = (uint32_t)ms_to_samples(src->time_ms, srate, nullptr);
	u.type = src->type;
	u.flags = src->flags;
	return u;
}

bool Engine::flush_updates(std::vector<OpUpdate> &batch, std::vector<uint8_t> &touched,
		std::string &err) {
	if (batch.empty()) return true;
	bool ok = backend_->apply_updates(batch.data(), batch.size(), err);
	for (const OpUpdate &u : batch) touched[u.op] = 0;
	batch.clear();
	return ok;
}

/* generator.c:348-377 (handle_event) with 245-343 mirrored for what the host
 * needs: operator times, modulator lists, pan line, wave ids. */
bool Engine::handle_event(Stream &st, const EventNode &e, std::vector<OpUpdate> &batch,
		std::vector<uint8_t> &touched, std::string &err) {
	const sauProgramEvent *pe = e.pe;
	for (size_t i = 0; i < pe->op_data_count; ++i) {
		const sauProgramOpData *od = &pe->op_data[i];
		if (od->id >= st.ops.size()) { err = "event refers to an operator id out of range"; return false; }
		OpMirror &m = st.ops[od->id];
		uint32_t gid = st.op_base + od->id;
		if (touched[gid] && !flush_updates(batch, touched, err)) return false;
		OpUpdate u;
		memset(&u, 0, sizeof u);
		u.op = gid;
		u.params = od->params;
		/* (an operator is of the type it was made with: the reference's update_op switches on the datum's type, generator.c:
		 * 283-343, which for a program whose later data name another type than the first -- no parser emits one; the fuzz of
		 * program images does -- reinterprets the node's union; here the later datum is applied as one of the operator's own type,
		 * so that the wave / noise / line id in its record stays one of that type's: found under ASan in round 5) */
		u.type = m.inited ? m.type : od->type;
		u.first = !m.inited;
		u.coeff = (float)(0x1p32 / (double)srate_);
		if (!m.inited) {
			m = OpMirror();
			m.inited = true;
			m.type = od->type;
			for (int k = 0; k < SAU_POP_NAMED; ++k) m.mods[k] = &g_no_ids;
			m.wave = SAU_WAVE_N_sin;
		}
		const bool is_osc = (u.type == SAU_POPT_N_wave || u.type == SAU_POPT_N_raseg);
		u.mode_main = od->mode.main;
		if (u.type == SAU_POPT_N_raseg) {
			u.ras_line = od->mode.ras.line;
			u.ras_flags = od->mode.ras.flags;
			u.ras_func = od->mode.ras.func;
			u.ras_level = od->mode.ras.level;
			u.ras_alpha = od->mode.ras.alpha;
			if (od->mode.ras.line == SAU_LINE_N_cub) m.ras_cub_seen = true;
			m.ras_kind = (uint32_t)od->mode.ras.line | ((uint32_t)od->mode.ras.func << 8) | ((uint32_t)od->mode.ras.flags << 16);
		}
		if (u.type == SAU_POPT_N_wave && (od->params & SAU_POPP_MODE))
			m.wave = od->mode.main < SAU_WAVE_NAMED ? od->mode.main : 0;
		if (u.type == SAU_POPT_N_noise && (od->params & SAU_POPP_MODE))
			m.wave = od->mode.main;
		u.phase = od->phase;
		u.seed = od->seed;
		const sauLine *src[L_COUNT] = {od->pan, od->amp, od->amp2, nullptr, nullptr, nullptr};
		if (is_osc) {
			src[L_FREQ] = od->freq; src[L_FREQ2] = od->freq2; src[L_PMA] = od->pm_a;
			if (od->fmods) m.mods[SAU_POP_N_fmod] = od->fmods;
			if (od->rfmods) m.mods[SAU_POP_N_rfmod] = od->rfmods;
			if (od->pmods) m.mods[SAU_POP_N_pmod] = od->pmods;
			if (od->apmods) m.mods[SAU_POP_N_apmod] = od->apmods;
			if (od->fpmods) m.mods[SAU_POP_N_fpmod] = od->fpmods;
		}
		for (uint32_t l = 0; l < L_COUNT; ++l) {
			u.line[l] = make_line_update(src[l], srate_);
			if (src[l]) m.line_set |= (uint8_t)(1u << l);
			if (src[l] && (src[l]->flags & SAU_LINEP_GOAL)) {
				m.goal_seen = true;
				if (l == L_FREQ || l == L_FREQ2) m.freq_goal_seen = true;
			}
			if (src[l] && (l == L_FREQ || l == L_FREQ2) && (src[l]->flags & (SAU_LINEP_STATE_RATIO | SAU_LINEP_GOAL_RATIO)))
				m.freq_ratio_seen = true;
		}
		u.loop_tails = loop_tails_ ? 1u : 0u;
		line_copy(m.pan, u.line[L_PAN], loop_tails_);
		if (od->params & SAU_POPP_TIME) {
			if (od->time.flags & SAU_TIMEP_IMPLICIT) {
				m.time = 0; m.time_inf = true;
				u.time = 0; u.time_inf = 1;
			} else {
				m.time = (uint32_t)ms_to_samples(od->time.v_ms, srate_, nullptr);
				m.time_inf = false;
				u.time = m.time; u.time_inf = 0;
			}
		}
		if (od->camods) m.mods[SAU_POP_N_camod] = od->camods;
		if (od->amods) m.mods[SAU_POP_N_amod] = od->amods;
		if (od->ramods) m.mods[SAU_POP_N_ramod] = od->ramods;
		batch.push_back(u);
		touched[gid] = 1;
	}
	if (pe->vo_id != SAU_PVO_NO_ID) {
		if (pe->vo_id >= st.voices.size()) { err = "event refers to a voice id out of range"; return false; }
		VoiceHost &vn = st.voices[pe->vo_id];
		vn.carr_op = pe->carr_op_id;
		vn.init = true;
		if (st.voice > pe->vo_id)
			st.voice = pe->vo_id;
		/* generator.c:233-240 */
		vn.duration = (vn.carr_op < st.ops.size()) ? st.ops[vn.carr_op].time : 0;
		vn.plan_valid = false;
		plans_dirty_ = true;
	} else if (pe->op_data_count > 0) {
		/* operators changed without a voice: graphs may have changed */
		for (VoiceHost &vn : st.voices) vn.plan_valid = false;
		plans_dirty_ = true;
	}
	return true;
}

bool Engine::rebuild_plans(std::string &err) {
	++rebuilds_;
	all_steps_.clear();
	all_fast_ids_.clear();
	all_fast_ids_full_.clear();
	all_op_ids_.clear();
	for (Stream &st : streams_) {
		for (size_t v = 0; v < st.voices.size(); ++v) {
			VoiceHost &vn = st.voices[v];
			PlanRef &ref = plan_refs_[st.vo_base + v];
			ref = PlanRef{0, 0, 0, 0};
			if (!vn.init) continue;
			if (!vn.plan_valid) {
				vn.shape = -1;
				uint64_t wmask = 0;
				if (plan_cache_ && ++shape_mark_ != 0 &&
				    voice_plan_shape(st.ops, vn.carr_op, shape_tokens_, shape_ids_, shape_stamp_, shape_mark_, wmask)) {
					uint64_t h = 1469598103934665603ull;
					for (uint32_t t : shape_tokens_) { h ^= t; h *= 1099511628211ull; }
					auto range = shape_by_hash_.equal_range(h);
					for (auto it = range.first; it != range.second; ++it)
						if (shapes_[it->second].tokens == shape_tokens_) { vn.shape = (int32_t)it->second; break; }
					if (vn.shape < 0) {
						Shape sh;
						sh.tokens = shape_tokens_;
						if (compile_voice_plan(st.ops, vn.carr_op, sh.plan, err) && sh.plan.op_ids == shape_ids_) {
							vn.shape = (int32_t)shapes_.size();
							shape_by_hash_.emplace(h, (uint32_t)shapes_.size());
							shapes_.push_back(std::move(sh));
						} else {
							err.clear(); /* (compiled again, and reported, below) */
						}
					}
				}
				if (vn.shape >= 0) {
					/* everything but the operator ids and the wave tables in use is the shape's */
					const VoicePlan &sp = shapes_[vn.shape].plan;
					if (plan_check_) {
						VoicePlan chk;
						std::string e2;
						const bool ok = compile_voice_plan(st.ops, vn.carr_op, chk, e2);
						if (!ok || chk.op_ids != shape_ids_ || chk.steps.size() != sp.steps.size() ||
						    memcmp(chk.steps.data(), sp.steps.data(), sp.steps.size() * sizeof(Step)) != 0 ||
						    chk.wave_mask != wmask || chk.n_fast != sp.n_fast || chk.n_fast_full != sp.n_fast_full ||
						    chk.n_slots != sp.n_slots || chk.n_main != sp.n_main || chk.no_fast != sp.no_fast ||
						    chk.static_block != sp.static_block || chk.selfmod != sp.selfmod || chk.n_chain != sp.n_chain ||
						    chk.n_osc != sp.n_osc || chk.has_camods != sp.has_camods || chk.carr_local != sp.carr_local || chk.wide != sp.wide) {
							err = "plan cache: a cached plan differs from a fresh compile (SAU_AMD_PLAN_CHECK)";
							return false;
						}
					}
					vn.plan.steps.clear(); vn.plan.fast_ids.clear(); vn.plan.fast_ids_full.clear(); /* (the shape's are used) */
					vn.plan.op_ids = shape_ids_;
					vn.plan.carr_local = sp.carr_local; vn.plan.n_slots = sp.n_slots; vn.plan.n_main = sp.n_main;
					vn.plan.n_fast = sp.n_fast; vn.plan.n_fast_full = sp.n_fast_full; vn.plan.wave_mask = wmask;
					vn.plan.has_camods = sp.has_camods; vn.plan.no_fast = sp.no_fast; vn.plan.static_block = sp.static_block;
					vn.plan.selfmod = sp.selfmod; vn.plan.n_chain = sp.n_chain; vn.plan.n_osc = sp.n_osc; vn.plan.wide = sp.wide;
					vn.plan.ras_cub = false;
					for (uint32_t id : shape_ids_) if (st.ops[id].ras_cub_seen) vn.plan.ras_cub = true;
					vn.plan.n_steps = (uint32_t)sp.steps.size();
				} else if (!compile_voice_plan(st.ops, vn.carr_op, vn.plan, err)) {
					/* a voice whose carrier never got data stays silent */
					vn.plan.steps.clear();
					vn.plan.fast_ids.clear();
					vn.plan.fast_ids_full.clear();
					vn.plan.op_ids.clear();
					vn.plan.n_steps = 0;
					if (err != "voice carrier operator was never initialised")
						return false;
					err.clear();
				} else {
					vn.plan.n_steps = (uint32_t)vn.plan.steps.size();
				}
				vn.plan_valid = true;
				if (tune_env("SAU_AMD_PLAN_DUMP")) { /* (debugging aid: the voice's plan and its block buffers in the time-parallel numberings) */
					const VoicePlan &pl = vn.shape >= 0 ? shapes_[vn.shape].plan : vn.plan;
					fprintf(stderr, "[sau-amd] plan: voice %zu steps %zu n_fast %u n_fast_full %u\n", v, pl.steps.size(), vn.plan.n_fast, vn.plan.n_fast_full);
					for (size_t i = 0; i < pl.steps.size(); ++i) {
						const Step &q = pl.steps[i];
						const FastIds f = i < pl.fast_ids_full.size() ? pl.fast_ids_full[i] : FastIds();
						fprintf(stderr, "[sau-amd]   %2zu kind %u flags %#x op %u which %u out %u freq %u fmul %u pm %u amp %u | full ids: out %d pm %d amp %d aux %d freq %d fmul %d\n",
								i, q.kind, q.flags, q.op, q.which, q.out, q.freq, q.fmul, q.pm, q.amp,
								(int)(int8_t)f.out, (int)(int8_t)f.pm, (int)(int8_t)f.amp, (int)(int8_t)f.aux, (int)(int8_t)f.freq, (int)(int8_t)f.fmul);
					}
				}
			}
			const VoicePlan &steps_of = vn.shape >= 0 ? shapes_[vn.shape].plan : vn.plan;
			ref.ops_ofs = (uint32_t)all_op_ids_.size();
			ref.nops = (uint32_t)vn.plan.op_ids.size();
			ref.plan_len = (uint32_t)steps_of.steps.size();
			if (vn.shape >= 0 && shapes_[vn.shape].placed == rebuilds_) {
				ref.plan_ofs = shapes_[vn.shape].plan_ofs; /* the shape's steps are in this upload already */
			} else {
				ref.plan_ofs = (uint32_t)all_steps_.size();
				if (vn.shape >= 0) { shapes_[vn.shape].plan_ofs = ref.plan_ofs; shapes_[vn.shape].placed = rebuilds_; }
				all_steps_.insert(all_steps_.end(), steps_of.steps.begin(), steps_of.steps.end());
				all_fast_ids_.insert(all_fast_ids_.end(), steps_of.fast_ids.begin(), steps_of.fast_ids.end());
				all_fast_ids_.resize(all_steps_.size());
				all_fast_ids_full_.insert(all_fast_ids_full_.end(), steps_of.fast_ids_full.begin(), steps_of.fast_ids_full.end());
				all_fast_ids_full_.resize(all_steps_.size());
			}
			for (uint32_t id : vn.plan.op_ids)
				all_op_ids_.push_back(st.op_base + id);
		}
	}
	/* An operator reachable from two voices would be advanced concurrently by
	 * two workgroups; the reference advances it twice in voice order. */
	{
		std::vector<uint8_t> seen(total_ops_, 0);
		for (Stream &st : streams_)
			for (VoiceHost &vn : st.voices) {
				if (!vn.init || vn.duration == 0) continue;
				for (uint32_t id : vn.plan.op_ids) {
					if (seen[st.op_base + id]) {
						err = "an operator is shared between two live voices (unsupported)";
						return false;
					}
					seen[st.op_base + id] = 1;
				}
			}
	}
	plans_dirty_ = false;
	all_fast_ids_.insert(all_fast_ids_.end(), all_fast_ids_full_.begin(), all_fast_ids_full_.end());
	return backend_->upload_plans(all_steps_.data(), all_fast_ids_.data(), all_steps_.size(),
			all_op_ids_.data(), all_op_ids_.size(), err);
}

/* How deep running sums may nest in this operator's subtree (see fast_kernel's running sums):
 * an oscillator whose frequency varies is a running sum one level above the deepest such
 * oscillator its frequency depends on. Conservative: a frequency sweep ever given counts as still
 * running, every child as a ratio child. Returns the level of what the operator puts out; `need`
 * collects the deepest oscillator level. The device takes the exact decision and falls back to
 * one wave in order when a voice is deeper than the passes launched. */
uint32_t Engine::estimate_sum_levels(const Stream &st, uint32_t op, uint32_t parent_dep, bool parent_varies,
		uint32_t &need, uint32_t nest) const {
	if (op >= st.ops.size() || nest > MAX_NEST) return 0;
	const OpMirror &m = st.ops[op];
	auto count = [](const sauProgramIDArr *a) { return a ? a->count : 0u; };
	const sauProgramIDArr *fm = m.mods[SAU_POP_N_fmod], *rfm = m.mods[SAU_POP_N_rfmod];
	uint32_t dep = parent_dep;
	const bool varies = parent_varies || m.freq_goal_seen || count(fm) || count(rfm);
	uint32_t out = 0;
	for (const sauProgramIDArr *lst : {fm, rfm})
		for (uint32_t i = 0; i < count(lst); ++i) {
			const uint32_t o = estimate_sum_levels(st, lst->ids[i], parent_dep, parent_varies || m.freq_goal_seen, need, nest + 1);
			if (o > dep) dep = o;
			if (o > out) out = o;
		}
	const bool osc = m.type == SAU_POPT_N_wave || m.type == SAU_POPT_N_raseg;
	const uint32_t own = (osc && varies) ? 1 + dep : 0;
	if (own > need) need = own;
	if (own > out) out = own;
	for (int use = 1; use < SAU_POP_NAMED; ++use) {
		if (use == SAU_POP_N_fmod || use == SAU_POP_N_rfmod) continue;
		const sauProgramIDArr *lst = m.mods[use];
		for (uint32_t i = 0; i < count(lst); ++i) {
			const uint32_t o = estimate_sum_levels(st, lst->ids[i], dep, varies, need, nest + 1);
			if (o > out) out = o;
		}
	}
	return out;
}

/* generator.c:854-878 (run_for_time) + 833-846 (run_voice), one segment. */
bool Engine::render_segment(uint32_t len, uint32_t offset, bool stereo, std::string &err) {
	std::vector<VoiceDesc> descs;
	std::vector<SegmentDesc::Stream> sdescs(streams_.size());
	uint32_t n_main = 1, n_fpool = 0, max_ops = 1, n_pan = 0, max_steps = 1, n_fast = 1, n_fast_full = 0;
	uint64_t wave_mask = 0;
	bool maybe_block = false, serial = false, may_scan = false, maybe_cub = false;
	uint32_t sum_levels = 0, n_chain_rows = 0, n_chain_slots = 0, n_inc_rows = 0, n_look_rows = 0, n_may_scan = 0;
	bool chain_rows_padded = false;
	for (size_t s = 0; s < streams_.size(); ++s) {
		Stream &st = streams_[s];
		SegmentDesc::Stream &sd = sdescs[s];
		sd.first_voice = (uint32_t)descs.size();
		sd.amp_scale = st.amp_scale;
		sd.write_len = 0;
		/* the reference's spans: from a call start or this program's latest event, to the call's
		 * end or its next event (generator.c:917-946) */
		Lattice lat;
		lat.call_len = lat_call_;
		lat.e0 = st.since_event < seg_call_pos_ ? (uint32_t)st.since_event : seg_call_pos_;
		lat.span_left = lat_call_ - seg_call_pos_;
		uint32_t ev_left = 0xffffffffu;
		if (st.event < st.events.size()) {
			/* (event_pos was advanced past this segment already) */
			const uint32_t wt = st.events[st.event].wait - (st.event_pos - len);
			if (wt < lat.span_left) lat.span_left = wt;
			ev_left = wt;
		}
		st.since_event += len;
		for (uint32_t v = st.voice; v < st.voices.size(); ++v) {
			VoiceHost &vn = st.voices[v];
			if (vn.duration == 0) continue;
			uint32_t run_len = std::min(vn.duration, len);
			vn.duration -= run_len;
			if (vn.carr_op >= st.ops.size()) continue;
			OpMirror &carr = st.ops[vn.carr_op];
			uint32_t out_len = 0;
			if (carr.time > 0) /* generator.c:839; implicit-time carriers have time 0 */
				out_len = carr.time_inf ? run_len : std::min(run_len, carr.time);
			if (out_len == 0 || vn.plan.n_steps == 0) continue;
			if (!carr.time_inf) carr.time -= out_len;
			const PlanRef &ref = plan_refs_[st.vo_base + v];
			VoiceDesc d;
			d.plan_ofs = ref.plan_ofs; d.plan_len = ref.plan_len;
			d.ops_ofs = ref.ops_ofs; d.nops = ref.nops;
			d.carr_local = vn.plan.carr_local;
			d.run_len = run_len;
			d.out_row = (uint32_t)descs.size();
			/* generator.c:756-762: dynamic pan needs per-sample values */
			bool dyn = (carr.pan.flags & LP_GOAL) || vn.plan.has_camods;
			d.pan_dynamic_row = dyn ? n_pan++ : ~0u;
			d.flags = (vn.plan.no_fast ? VD_NO_FAST : 0) | (vn.duration ? VD_MORE : 0) | (loop_tails_ ? VD_TAILS : 0) |
				(vn.plan.wide ? VD_WIDE : 0);
			d.lat = lat;
			d.ev_left = ev_left;
			d.chain_base = n_chain_rows; d.chain_slot = n_chain_rows; d.n_chain = vn.plan.n_chain;
			n_chain_rows += vn.plan.n_chain;
			d.inc_base = 0; d.n_inc = 0; /* (set below for voices that may have running-sum phases) */
			d.look_base = 0; d.n_look = 0;
			if (dyn) line_begin(carr.pan, out_len, false, 0.f, lat, 0);
			else line_skip(carr.pan, out_len, lat, 0);
			descs.push_back(d);
			if (out_len > sd.write_len) sd.write_len = out_len;
			n_main = std::max(n_main, vn.plan.n_main);
			if (!vn.plan.no_fast) n_fast = std::max(n_fast, vn.plan.n_fast);
			n_fpool = std::max(n_fpool, vn.plan.n_slots - vn.plan.n_main);
			max_ops = std::max(max_ops, (uint32_t)vn.plan.op_ids.size());
			max_steps = std::max(max_steps, vn.plan.n_steps);
			wave_mask |= vn.plan.wave_mask;
			/* will the time-parallel path surely cover this voice's whole run? */
			bool voice_block = vn.plan.static_block; /* this voice may leave the closed-form path */
			if (vn.plan.static_block || vn.plan.no_fast) maybe_block = true;
			if (vn.plan.selfmod) serial = true;
			if (vn.plan.ras_cub && loop_tails_) maybe_cub = true;
			for (uint32_t id : vn.plan.op_ids) {
				OpMirror &m = st.ops[id];
				if (m.goal_seen || (m.line_set & (1u << L_PMA))) { maybe_block = true; voice_block = true; }
				if (m.line_set & (1u << L_PMA)) serial = true;
				if (id != vn.carr_op && !m.time_inf) {
					/* conservative mirror: non-carriers tick whenever the voice runs */
					if (m.time < run_len) maybe_block = true;
					m.time -= std::min(m.time, out_len);
				}
			}
			if (out_len < run_len) maybe_block = true;
			if (voice_block && !vn.plan.no_fast) {
				n_fast_full = std::max(n_fast_full, vn.plan.n_fast_full);
				may_scan = true;
				++n_may_scan;
				descs.back().inc_base = n_inc_rows; descs.back().n_inc = vn.plan.n_osc;
				n_inc_rows += vn.plan.n_osc;
				if (vn.plan.n_chain == 0) { /* single-pass running sums: one look-back row per oscillator, eight at most */
					descs.back().look_base = n_look_rows; descs.back().n_look = std::min<uint32_t>(vn.plan.n_osc, 8);
					n_look_rows += descs.back().n_look;
				}
				if (sum_levels < 3) (void)estimate_sum_levels(st, vn.carr_op, 0, false, sum_levels, 0);
			}
		}
		sd.n_voices = (uint32_t)descs.size() - sd.first_voice;
		if (sd.write_len > 0) {
			size_t end = (size_t)(offset - (uint32_t)st.part_start) + sd.write_len;
			if (end > st.part_gen) st.part_gen = end;
		}
	}
	/* Every frame of the call belongs to some segment, and a segment's mixer writes a stream's frames [0, write_len): what lies
	 * behind -- the stream's last voice has ended, or nothing of it sounds in this segment -- is cleared here, runs of streams
	 * with the same share at a time (generator.c:911-914 clears the caller's whole buffer ahead of every call and adds into it;
	 * round 6: a batch of 64 one-minute renders cleared 339 MB per run that its mixers then wrote frame for frame) */
	for (size_t s = 0; s < sdescs.size();) {
		size_t e = s + 1;
		while (e < sdescs.size() && sdescs[e].write_len == sdescs[s].write_len) ++e;
		if (sdescs[s].write_len < len &&
		    !backend_->zero_pcm((uint32_t)s, (uint32_t)(e - s), offset + sdescs[s].write_len, len - sdescs[s].write_len, stereo, err))
			return false;
		s = e;
	}
	if (descs.empty()) return true;
	/* Feedback chains are dealt to the chain kernels' waves 64 rows at a time (k_chain.h), and a wave whose R chains agree in
	 * line shape, function and flags takes the copy of the loop with those dispatches scalar -- 340 ns per frame against about
	 * 1000 where the 64 differ and every kind present is evaluated for all. So the rows of a
	 * segment with R feedback are numbered kind by kind (voices without an R operator first, in voice order; nothing else
	 * depends on the numbering: a row belongs to its voice through chain_base, a lane through chain_slot). */
	static const bool no_chain_sort = tune_env("SAU_AMD_NO_CHAIN_SORT") != nullptr; /* (A/B: rows in voice order) */
	if (n_chain_rows > 1 && !no_chain_sort) { /* (until round 6: from 65 rows on; eight voices of eight kinds in one wave took 1086 ns per frame) */
		std::vector<std::pair<uint32_t, uint32_t>> order; /* (kind, index into descs) of the voices with chain rows */
		bool any_r = false;
		uint32_t vi = 0;
		for (size_t s = 0; s < streams_.size(); ++s) {
			const Stream &st = streams_[s];
			for (; vi < sdescs[s].first_voice + sdescs[s].n_voices; ++vi) {
				if (!descs[vi].n_chain) continue;
				uint32_t kind = 0;
				for (uint32_t k = 0; k < descs[vi].nops; ++k) {
					const OpMirror &m = st.ops[all_op_ids_[descs[vi].ops_ofs + k] - st.op_base];
					/* (... and whether its lines were ever given sweeps: the feeder waves' line evaluations dispatch on the shapes) */
					if (m.type == SAU_POPT_N_raseg) { kind = 1u + (m.ras_kind | (m.freq_goal_seen ? 1u << 24 : 0u) | (m.goal_seen ? 1u << 25 : 0u)); any_r = true; break; }
				}
				order.emplace_back(kind, vi);
			}
		}
		if (any_r) {
			std::stable_sort(order.begin(), order.end(), [](const std::pair<uint32_t, uint32_t> &a, const std::pair<uint32_t, uint32_t> &b) { return a.first < b.first; });
			/* ... and every kind begins a wave of its own: the LANES are numbered with gaps (a multiple of 64 per kind; the lanes in
			 * between stay unused: ChainDesc.n = 0), the rows without -- a gap costs a descriptor, no memory -- while the waves, one
			 * workgroup each, still run side by side (five workgroups a CU by their LDS: 1280; the cap is 1024). None then holds two
			 * kinds (1024 voices of 24 kinds: 861 ns per frame numbered kind by kind, 1056 in voice order, profiles/r06_ab.txt;
			 * until the rows and the lanes had numbers of their own the gaps cost rows, and 4096 voices x 10 s of 96 kinds did not
			 * fit half the budget: 564 ns per frame where the slowest kind alone takes 405) */
			uint32_t padded = 0, prev = ~0u;
			for (const auto &o : order) {
				if (o.first != prev && o.first != 0 && prev != ~0u) padded = (padded + 63u) & ~63u;
				prev = o.first;
				padded += descs[o.second].n_chain;
			}
			const bool pad = padded <= 64u * 1024u || padded <= 2 * n_chain_rows;
			uint32_t at = 0, slot = 0;
			prev = ~0u;
			for (const auto &o : order) {
				if (pad && o.first != prev && o.first != 0 && prev != ~0u) slot = (slot + 63u) & ~63u;
				prev = o.first;
				descs[o.second].chain_base = at; at += descs[o.second].n_chain;
				descs[o.second].chain_slot = slot; slot += descs[o.second].n_chain;
			}
			if (pad && slot != at) { n_chain_slots = slot; chain_rows_padded = true; }
		}
	}
	SegmentDesc seg;
	seg.len = len; seg.pcm_offset = offset; seg.stereo = stereo; seg.swap_bytes = pcm_swap_;
	seg.voices = descs.data(); seg.n_voices = (uint32_t)descs.size();
	seg.streams = sdescs.data(); seg.n_streams = (uint32_t)sdescs.size();
	seg.n_slots = n_main + n_fpool; seg.n_main = n_main; seg.n_fast = n_fast; seg.n_fast_full = n_fast_full; seg.max_ops = max_ops; seg.n_pan_rows = n_pan;
	seg.max_steps = max_steps;
	seg.wave_mask = wave_mask;
	seg.maybe_block = maybe_block;
	seg.serial = serial;
	seg.maybe_cub = maybe_cub;
	seg.may_scan = may_scan;
	seg.sum_levels = sum_levels;
	seg.n_chain_rows = n_chain_rows;
	seg.n_chain_slots = n_chain_slots > n_chain_rows ? n_chain_slots : n_chain_rows;
	seg.n_inc_rows = n_inc_rows;
	seg.n_look_rows = n_look_rows;
	seg.n_may_scan = n_may_scan;
	seg.chain_rows_padded = chain_rows_padded;
	return backend_->render(seg, err);
}

bool Engine::reserve(size_t frames, bool stereo, std::string &err) {
	if (frames > UINT32_MAX) { err = "buffer too long"; return false; }
	if ((uint32_t)frames > reserved_frames_ || (stereo && !reserved_stereo_)) {
		const uint32_t want = std::max((uint32_t)frames, reserved_frames_);
		if (!backend_->reserve_frames(want, stereo || reserved_stereo_, err)) return false;
		reserved_frames_ = want;
		reserved_stereo_ = stereo || reserved_stereo_;
	}
	return true;
}

bool Engine::snapshot(Snapshot &s, int slot, std::string &err) {
	s.valid = false;
	s.streams.resize(streams_.size());
	for (size_t i = 0; i < streams_.size(); ++i) {
		const Stream &st = streams_[i];
		Snapshot::StreamState &d = s.streams[i];
		d.event = st.event; d.event_pos = st.event_pos; d.voice = st.voice; d.since_event = st.since_event;
		d.voices.resize(st.voices.size());
		for (size_t v = 0; v < st.voices.size(); ++v)
			d.voices[v] = Snapshot::StreamState::Vo{st.voices[v].duration, st.voices[v].carr_op, st.voices[v].init};
		d.ops = st.ops;
	}
	s.frames_done = frames_done_;
	s.call_len = call_len_; s.lat_call = lat_call_; s.call_phase = call_phase_;
	if (!backend_->save_state(slot, err)) return false;
	s.valid = true;
	return true;
}

bool Engine::restore(const Snapshot &s, int slot, std::string &err) {
	if (!s.valid || s.streams.size() != streams_.size()) { err = "no snapshot to go back to"; return false; }
	for (size_t i = 0; i < streams_.size(); ++i) {
		Stream &st = streams_[i];
		const Snapshot::StreamState &d = s.streams[i];
		st.event = d.event; st.event_pos = d.event_pos; st.voice = d.voice; st.since_event = d.since_event;
		st.call_gen = st.part_start = st.part_gen = 0;
		for (size_t v = 0; v < st.voices.size(); ++v) {
			VoiceHost &vn = st.voices[v];
			vn.duration = d.voices[v].duration; vn.carr_op = d.voices[v].carr_op; vn.init = d.voices[v].init;
			vn.plan_valid = false; /* (compiled again from the mirrors below: the graphs may have been other ones then) */
		}
		st.ops = d.ops;
	}
	plans_dirty_ = true;
	frames_done_ = s.frames_done;
	call_len_ = s.call_len; lat_call_ = s.lat_call; call_phase_ = s.call_phase;
	return backend_->load_state(slot, err);
}

/* generator.c:905-973, for all streams in lock step. */
bool Engine::run(int16_t *const *host_bufs, size_t buf_len, bool stereo,
		bool *more, size_t *out_len, std::string &err) {
	if (buf_len > UINT32_MAX) { err = "buffer too long"; return false; }
	const uint32_t total = (uint32_t)buf_len;
	if (total > reserved_frames_ || (stereo && !reserved_stereo_)) {
		uint32_t want = std::max(total, reserved_frames_);
		if (!backend_->reserve_frames(want, stereo || reserved_stereo_, err)) return false;
		reserved_frames_ = want;
		reserved_stereo_ = stereo || reserved_stereo_;
	}
	/* (generator.c:911-914, the buffer cleared ahead of the call: render_segment clears what the mixers leave) */
	for (Stream &st : streams_) { st.call_gen = 0; st.part_start = 0; st.part_gen = 0; }
	{
		const uint32_t want = call_len_ ? call_len_ : (total ? total : 1u);
		if (want != lat_call_ || !call_len_) { lat_call_ = want; call_phase_ = 0; }
	}
	std::vector<OpUpdate> batch;
	std::vector<uint8_t> touched(total_ops_, 0);
	uint32_t pos = 0, remaining = total;
	while (remaining > 0) {
		const auto t_loop = std::chrono::steady_clock::now();
		for (Stream &st : streams_) {
			bool split = false;
			while (st.event < st.events.size()) {
				const EventNode &e = st.events[st.event];
				if (st.event_pos < e.wait) break;
				if (!split && pos > (uint32_t)st.part_start) {
					/* this stream's own split point: generator.c:939-946 */
					st.call_gen += pos - st.part_start;
					st.part_start = pos; st.part_gen = 0;
				}
				split = true;
				if (!handle_event(st, e, batch, touched, err)) return false;
				++st.event;
				st.event_pos = 0;
				st.since_event = 0;
			}
		}
		const bool trace = getenv("SAU_AMD_DEBUG_CREATE") != nullptr;
		const auto t_ev = std::chrono::steady_clock::now();
		if (trace && std::chrono::duration<double, std::milli>(t_ev - t_loop).count() > 0.05)
			fprintf(stderr, "saugns_amd: segment at %u: events applied on the host in %.3f ms\n", pos,
					std::chrono::duration<double, std::milli>(t_ev - t_loop).count());
		if (!flush_updates(batch, touched, err)) return false;
		const auto t_fl = std::chrono::steady_clock::now();
		const bool was_dirty = plans_dirty_;
		if (plans_dirty_ && !rebuild_plans(err)) return false;
		if (trace && was_dirty)
			fprintf(stderr, "saugns_amd: segment at %u: flush of operator updates %.3f ms, plans %.3f ms\n", pos,
					std::chrono::duration<double, std::milli>(t_fl - t_ev).count(),
					std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t_fl).count());
		uint32_t seg = remaining;
		for (Stream &st : streams_) {
			if (st.event < st.events.size()) {
				uint32_t wt = st.events[st.event].wait - st.event_pos;
				if (wt < seg) seg = wt;
			}
		}
		/* An operator that runs out of time inside a segment leaves its voice to the block loop
		 * from that frame to the segment's end (the time-parallel path renders a voice up to its
		 * first such frame); in the next segment it is simply left out. So a long segment ends
		 * soon after the earliest such frame -- on a grid, so that staggered envelopes cost a
		 * handful of segments per run, not one each. (The mirrored times are conservative: an
		 * operator may in fact stop earlier, which costs one superfluous cut.) */
		/* (SAU_AMD_EXPIRY_GRID=<frames>, a tuning switch: 1 cuts at the frame itself -- no voice then leaves the time-parallel path
		 * for an operator that runs out, at a segment per such frame; measured over the corpus in DESIGN.md 10) */
		static const uint32_t EXPIRY_GRID = [] { const char *e = tune_env("SAU_AMD_EXPIRY_GRID"); const long n = e ? atol(e) : 0; return n >= 1 ? (uint32_t)n : 8192u; }();
		if (seg > EXPIRY_GRID) {
			uint32_t first = seg;
			size_t chains = 0;
			for (Stream &st : streams_)
				for (uint32_t v = st.voice; v < st.voices.size(); ++v) {
					const VoiceHost &vn = st.voices[v];
					if (vn.duration == 0 || vn.carr_op >= st.ops.size()) continue;
					chains += vn.plan.n_chain;
					/* (voices that merely have running-sum phases need no cap: they take one pass with look-back, whose
					 * state is per row group; where they take several passes instead, the backend saves increments only
					 * for segments within CHAIN_SEG and recomputes them beyond) */
					for (uint32_t id : vn.plan.op_ids) {
						const OpMirror &m = st.ops[id];
						if (id != vn.carr_op && !m.time_inf && m.time > 0 && m.time < first) first = m.time;
					}
				}
			if (first < seg) {
				const uint32_t cut = (first + EXPIRY_GRID - 1) / EXPIRY_GRID * EXPIRY_GRID;
				if (cut < seg) seg = cut;
			}
			if (chains && seg > chain_seg_frames(chains, backend_->chain_rows_budget())) seg = chain_seg_frames(chains, backend_->chain_rows_budget()); /* rows in HBM carry one segment of every recurrence */
		}
		for (Stream &st : streams_)
			if (st.event < st.events.size()) st.event_pos += seg;
		seg_call_pos_ = (uint32_t)(((uint64_t)call_phase_ + pos) % lat_call_);
		if (!render_segment(seg, pos, stereo, err)) return false;
		pos += seg;
		remaining -= seg;
	}
	frames_done_ += total;
	call_phase_ = (uint32_t)(((uint64_t)call_phase_ + total) % lat_call_);
	for (size_t s = 0; s < streams_.size(); ++s) {
		Stream &st = streams_[s];
		st.call_gen += st.part_gen;
		bool ended = false;
		for (;;) { /* generator.c:953-967 */
			if (st.voice == st.voices.size()) {
				if (st.event != st.events.size()) break;
				ended = true;
				break;
			}
			if (st.voices[st.voice].duration != 0) break;
			++st.voice;
		}
		if (more) more[s] = !ended;
		if (out_len) out_len[s] = ended ? st.call_gen : buf_len;
		if (host_bufs && host_bufs[s]) {
			if (!backend_->fetch_pcm((uint32_t)s, host_bufs[s], total, stereo, err))
				return false;
		}
	}
	return true;
}

} /* namespace sauengine */
