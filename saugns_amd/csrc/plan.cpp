/* plan.cpp -- flatten one voice's operator graph into a step list.
 *
 * The reference evaluates a voice by recursion (generator.c:675-729 run_block,
 * 548-664 run_block_wosc/rasg, 505-541 amp/noise, 448-498 parameter helpers,
 * 749-788 mix_add).  Here the same post-order walk is done once per graph
 * change on the host and recorded as steps over numbered block buffers
 * ("slots"); the device then replays the steps for every block.  Buffer
 * numbers are allocated by liveness instead of the reference's fixed
 * 7-per-nesting-level layout, and the phase buffer is never materialised
 * (phase accumulation, lookup and mixing are fused into ST_OSC).
 */
#include "engine.h"
#include <unordered_map>

namespace sauengine {

namespace {

/* ids a wide plan may use per pool (memory index = n_main + frequency id stays below 16 bits) */
constexpr uint32_t WIDE_MAIN_MAX = 4096, WIDE_FREQ_MAX = 4095;

struct Compiler {
	const std::vector<OpMirror> &ops;
	VoicePlan &out;
	std::string &err;
	std::vector<uint32_t> path;     /* operators being evaluated (cycle guard) */
	std::unordered_map<uint32_t, uint32_t> local;
	/* buffer ids as the compiler hands them out: 16 bits (WideStep), main pool 1.., frequency pool from
	 * WSLOT_FBASE; plans whose ids fit a Step's 8 bits are written narrow at the end (finish()) */
	std::vector<WideStep> wsteps;
	std::vector<bool> used = std::vector<bool>(0x10000, false);
	uint32_t high = 0;       /* highest main-pool slot */
	uint32_t high_f = 0;     /* number of frequency-pool slots */
	bool failed = false;

	Compiler(const std::vector<OpMirror> &o, VoicePlan &p, std::string &e)
		: ops(o), out(p), err(e) { used[SCRATCH_SLOT] = true; }

	uint16_t alloc() {
		for (uint32_t s = 1; s < WIDE_MAIN_MAX; ++s) {
			if (!used[s]) {
				used[s] = true;
				if (s > high) high = s;
				return (uint16_t)s;
			}
		}
		if (!failed) { failed = true; err = "operator graph needs more than 4095 block buffers"; }
		return 1;
	}
	/* Frequency blocks get numbers from a second pool: while an operator's frequency is one value
	 * they are never materialised, so a launch that only runs such voices needs no memory for them. */
	uint16_t alloc_f() {
		for (uint32_t s = WSLOT_FBASE; s < WSLOT_FBASE + WIDE_FREQ_MAX; ++s) {
			if (!used[s]) {
				used[s] = true;
				if (s - WSLOT_FBASE + 1 > high_f) high_f = s - WSLOT_FBASE + 1;
				return (uint16_t)s;
			}
		}
		if (!failed) { failed = true; err = "operator graph needs more than 4095 frequency buffers"; }
		return WSLOT_FBASE;
	}
	void release(uint16_t s) { if (s != NO_WSLOT) used[s] = false; }

	uint32_t local_of(uint32_t op) {
		auto it = local.find(op);
		if (it != local.end()) return it->second;
		uint32_t idx = (uint32_t)out.op_ids.size();
		out.op_ids.push_back(op);
		local[op] = idx;
		return idx;
	}

	WideStep &emit(uint8_t kind, uint32_t op_local) {
		WideStep s;
		s.kind = kind; s.flags = 0;
		s.out = s.freq = s.fmul = s.pm = s.fpm = s.amp = s.sm = NO_WSLOT;
		s.which = 0; s.tmp = NO_WSLOT; s.prov = NO_WSLOT;
		s.op = op_local;
		wsteps.push_back(s);
		return wsteps.back();
	}

	static uint32_t count(const sauProgramIDArr *a) { return a ? a->count : 0; }

	void children(const sauProgramIDArr *ids, uint16_t dst, uint16_t freq, uint16_t prov,
			bool wave_env, bool layer_all) {
		for (uint32_t i = 0; i < count(ids); ++i)
			eval(ids->ids[i], dst, freq, prov, wave_env, layer_all ? true : (i > 0), false);
	}

	/* parameter with optional range modulation + additive modulators into a
	 * slot: generator.c:448-477. Returns with `dst` holding the values. */
	void param_to_slot(uint32_t lop, uint32_t line, uint32_t line2, uint16_t dst,
			uint16_t mul, uint16_t mul_prov, uint16_t child_freq, uint16_t child_prov,
			const sauProgramIDArr *mods, const sauProgramIDArr *r_mods, bool &first) {
		WideStep &s = emit(ST_LINE, lop);
		s.which = (uint8_t)line; s.out = dst; s.fmul = mul; s.prov = mul_prov;
		s.tmp = (uint16_t)line2;
		if (first) { s.flags |= SF_BEGIN; first = false; }
		/* a frequency block that nothing adds into may stay a single value */
		if (line != L_FREQ || count(mods) || count(r_mods)) s.flags |= SF_FORCE;
		if (count(r_mods) == 0) {
			s.flags |= SF_SKIP2;
		} else {
			uint16_t r = alloc();
			WideStep &s2 = emit(ST_LINE, lop);
			s2.which = (uint8_t)line2; s2.out = r; s2.fmul = mul; s2.prov = mul_prov;
			s2.tmp = (uint16_t)line2; s2.flags |= SF_FORCE;
			uint16_t m = alloc();
			children(r_mods, m, child_freq, child_prov, true, false);
			WideStep &l = emit(ST_LERP, lop);
			l.out = dst; l.freq = r; l.pm = m;
			release(m); release(r);
		}
		if (count(mods) > 0)
			children(mods, dst, child_freq, child_prov, false, true);
	}

	/* One operator, combined into slot `dst`. Returns the slot that holds its
	 * frequency block when keep_freq (caller releases it), else NO_WSLOT. */
	uint16_t eval(uint32_t op, uint16_t dst, uint16_t parent_freq, uint16_t parent_prov,
			bool wave_env, bool layer, bool keep_freq) {
		if (failed) return NO_WSLOT;
		for (uint32_t p : path) {
			if (p == op) { /* generator.c:685-689 */
				WideStep &z = emit(ST_ZERO, local_of(op));
				z.out = dst;
				return NO_WSLOT;
			}
		}
		if (op >= ops.size() || !ops[op].inited) {
			if (!layer) { WideStep &z = emit(ST_ZERO, 0); z.out = dst; }
			return NO_WSLOT;
		}
		if (path.size() >= MAX_NEST) {
			failed = true; err = "operator nesting deeper than 256 levels";
			return NO_WSLOT;
		}
		const OpMirror &m = ops[op];
		const uint32_t lop = local_of(op);
		const uint16_t me = lop < NO_WSLOT ? (uint16_t)lop : NO_WSLOT;
		path.push_back(op);
		bool first = true;
		const bool is_osc = (m.type == SAU_POPT_N_wave || m.type == SAU_POPT_N_raseg);
		const sauProgramIDArr *amods = m.mods[SAU_POP_N_amod];
		const sauProgramIDArr *ramods = m.mods[SAU_POP_N_ramod];
		uint16_t F = NO_WSLOT, P = NO_WSLOT, Q = NO_WSLOT, A = NO_WSLOT, S = NO_WSLOT, T = NO_WSLOT;
		uint8_t osc_flags = 0;
		if (is_osc) {
			const sauProgramIDArr *fmods = m.mods[SAU_POP_N_fmod];
			const sauProgramIDArr *rfmods = m.mods[SAU_POP_N_rfmod];
			const sauProgramIDArr *pmods = m.mods[SAU_POP_N_pmod];
			const sauProgramIDArr *fpmods = m.mods[SAU_POP_N_fpmod];
			const sauProgramIDArr *apmods = m.mods[SAU_POP_N_apmod];
			bool any_child = count(fmods) || count(rfmods) || count(pmods) ||
				count(fpmods) || count(amods) || count(ramods) || count(apmods);
			if (any_child || keep_freq) {
				F = alloc_f();
				param_to_slot(lop, L_FREQ, L_FREQ2, F, parent_freq, parent_prov, F, me, fmods, rfmods, first);
			} else {
				osc_flags |= SF_SKIP_FREQ2;
			}
			if (count(pmods)) { P = alloc(); children(pmods, P, F, me, false, false); }
			if (count(fpmods)) { Q = alloc(); children(fpmods, Q, F, me, false, false); }
			if (count(amods) || count(ramods)) {
				A = alloc();
				param_to_slot(lop, L_AMP, L_AMP2, A, NO_WSLOT, NO_WSLOT, F, me, amods, ramods, first);
			} else {
				osc_flags |= SF_SKIP_AMP2;
			}
			if (count(apmods)) { /* generator.c:479-498 */
				S = alloc();
				WideStep &sm = emit(ST_SMLINE, lop);
				sm.out = S;
				if (first) { sm.flags |= SF_BEGIN; first = false; }
				children(apmods, S, F, me, false, true);
			} else if (m.line_set & (1u << L_PMA)) {
				osc_flags |= SF_SM_INLINE;
			}
			if (m.type == SAU_POPT_N_raseg && (S != NO_WSLOT || (osc_flags & SF_SM_INLINE)))
				T = alloc();
			if (m.type == SAU_POPT_N_wave)
				out.wave_mask |= 1ull << (m.wave & 63);
		} else {
			if (count(amods) || count(ramods)) {
				A = alloc();
				param_to_slot(lop, L_AMP, L_AMP2, A, NO_WSLOT, NO_WSLOT, NO_WSLOT, NO_WSLOT, amods, ramods, first);
			} else {
				osc_flags |= SF_SKIP_AMP2;
			}
		}
		WideStep &o = emit(ST_OSC, lop);
		o.out = dst; o.freq = F; o.fmul = parent_freq; o.pm = P; o.fpm = Q;
		o.prov = (F != NO_WSLOT) ? me : parent_prov;
		o.amp = A; o.sm = S; o.tmp = T;
		o.flags = osc_flags | SF_END;
		if (first) o.flags |= SF_BEGIN;
		if (wave_env) o.flags |= SF_WAVE_ENV;
		if (layer) o.flags |= SF_LAYER;
		release(T); release(S); release(A); release(Q); release(P);
		if (!keep_freq) { release(F); F = NO_WSLOT; }
		path.pop_back();
		return F;
	}

	/* The step list as the device reads it: 8-bit ids when they fit (same numbers as ever: lowest free
	 * first), else step pairs (sau_dev_types.h: wide plans). */
	void finish() {
		out.wide = high >= FSLOT_BASE || high_f > 250u - FSLOT_BASE;
		out.steps.clear();
		out.steps.reserve(wsteps.size() * (out.wide ? 2 : 1));
		auto lo_id = [&](uint16_t id) -> uint8_t {
			if (out.wide) return (uint8_t)(id & 0xFF);
			if (id == NO_WSLOT) return NO_SLOT;
			return (uint8_t)(id < WSLOT_FBASE ? id : FSLOT_BASE + (id - WSLOT_FBASE));
		};
		for (const WideStep &w : wsteps) {
			Step s;
			s.kind = w.kind; s.flags = w.flags; s.which = w.which; s.op = w.op;
			s.out = lo_id(w.out); s.freq = lo_id(w.freq); s.fmul = lo_id(w.fmul); s.pm = lo_id(w.pm);
			s.fpm = lo_id(w.fpm); s.amp = lo_id(w.amp); s.sm = lo_id(w.sm);
			s.tmp = w.kind == ST_OSC ? lo_id(w.tmp) : (uint8_t)w.tmp;
			if (out.wide) s.prov = (uint8_t)(w.prov & 0xFF);
			else s.prov = w.prov < 255 ? (uint8_t)w.prov : NO_SLOT;
			out.steps.push_back(s);
			if (out.wide) {
				Step h;
				h.kind = ST_WIDE; h.flags = 0; h.which = 0; h.op = w.op;
				h.out = (uint8_t)(w.out >> 8); h.freq = (uint8_t)(w.freq >> 8); h.fmul = (uint8_t)(w.fmul >> 8);
				h.pm = (uint8_t)(w.pm >> 8); h.fpm = (uint8_t)(w.fpm >> 8); h.amp = (uint8_t)(w.amp >> 8);
				h.sm = (uint8_t)(w.sm >> 8);
				h.tmp = w.kind == ST_OSC ? (uint8_t)(w.tmp >> 8) : 0;
				h.prov = (uint8_t)(w.prov >> 8);
				out.steps.push_back(h);
			}
		}
	}
};

} /* namespace */

/* The shape of a voice's operator graph as compile_voice_plan sees it: one token stream over the same walk
 * (local_of() order, the modulator lists in the order eval() visits them) holding everything the step list depends
 * on -- operator types, which lists have members, the pm_a line, red noise, pan modulators, a pan ramp pending.
 * Voices with equal streams get equal step lists; only their operator ids (and wave tables in use) differ. Banks
 * of thousands of like voices (BASELINE configs 2, 3, 5) then compile one plan, not thousands.
 * False: not cacheable (an operator met twice, a cycle, a member that never got data, very deep nesting) --
 * compile_voice_plan decides those. */
namespace {
struct ShapeWalk {
	const std::vector<OpMirror> &ops;
	std::vector<uint32_t> &tokens, &op_ids;
	std::vector<uint32_t> &stamp; /* per operator: the voice counter when it was last seen */
	uint32_t mark;
	uint64_t wave_mask = 0;
	uint32_t depth = 0;
	bool ok = true;
	static uint32_t count(const sauProgramIDArr *a) { return a ? a->count : 0; }
	void list(const sauProgramIDArr *ids) { for (uint32_t i = 0; ok && i < count(ids); ++i) walk(ids->ids[i]); }
	void walk(uint32_t op) {
		if (op >= ops.size() || !ops[op].inited || stamp[op] == mark || depth >= 64) { ok = false; return; }
		stamp[op] = mark;
		op_ids.push_back(op);
		const OpMirror &m = ops[op];
		const bool is_osc = m.type == SAU_POPT_N_wave || m.type == SAU_POPT_N_raseg;
		uint32_t t = m.type | ((m.line_set & (1u << L_PMA)) ? 0x100u : 0u) |
			((m.type == SAU_POPT_N_noise && m.wave == SAU_NOISE_N_re) ? 0x200u : 0u) |
			(m.freq_goal_seen ? 0x400u : 0u) | (m.goal_seen ? 0x800u : 0u) | (m.freq_ratio_seen ? 0x1000u : 0u); /* (which of its lines keep their buffers: fast_slot_compact's lean form) */
		for (int use = 1; use < SAU_POP_NAMED; ++use) if (count(m.mods[use])) t |= 0x1000u << use;
		tokens.push_back(t);
		for (int use = 1; use < SAU_POP_NAMED; ++use) if (count(m.mods[use])) tokens.push_back(count(m.mods[use]));
		if (m.type == SAU_POPT_N_wave) wave_mask |= 1ull << (m.wave & 63);
		++depth;
		if (is_osc) {
			list(m.mods[SAU_POP_N_rfmod]); list(m.mods[SAU_POP_N_fmod]);
			list(m.mods[SAU_POP_N_pmod]); list(m.mods[SAU_POP_N_fpmod]);
			list(m.mods[SAU_POP_N_ramod]); list(m.mods[SAU_POP_N_amod]);
			list(m.mods[SAU_POP_N_apmod]);
		} else {
			list(m.mods[SAU_POP_N_ramod]); list(m.mods[SAU_POP_N_amod]);
		}
		--depth;
	}
};
} /* namespace */

bool voice_plan_shape(const std::vector<OpMirror> &ops, uint32_t carrier, std::vector<uint32_t> &tokens,
		std::vector<uint32_t> &op_ids, std::vector<uint32_t> &stamp, uint32_t mark, uint64_t &wave_mask) {
	tokens.clear();
	op_ids.clear();
	if (carrier >= ops.size() || !ops[carrier].inited) return false;
	if (stamp.size() < ops.size()) stamp.resize(ops.size(), 0);
	ShapeWalk w{ops, tokens, op_ids, stamp, mark};
	const OpMirror &cm = ops[carrier];
	tokens.push_back((cm.pan.flags & LP_GOAL) ? 1u : 0u);
	w.walk(carrier);
	/* (the carrier's camods come after everything else: compile_voice_plan) -- they are in its token (use bits and
	 * counts); their members: */
	if (w.ok) w.list(cm.mods[SAU_POP_N_camod]);
	wave_mask = w.wave_mask;
	return w.ok;
}

bool compile_voice_plan(const std::vector<OpMirror> &ops, uint32_t carrier,
		VoicePlan &out, std::string &err) {
	out.steps.clear();
	out.op_ids.clear();
	out.wave_mask = 0;
	Compiler c(ops, out, err);
	if (carrier >= ops.size() || !ops[carrier].inited) {
		err = "voice carrier operator was never initialised";
		return false;
	}
	const OpMirror &cm = ops[carrier];
	const sauProgramIDArr *camods = cm.mods[SAU_POP_N_camod];
	out.has_camods = camods && camods->count > 0;
	out.carr_local = c.local_of(carrier);
	uint16_t V = c.alloc();
	/* generator.c:833-846 run_voice -> run_block(carrier, NULL, false, false) */
	uint16_t F = c.eval(carrier, V, NO_WSLOT, NO_WSLOT, false, false, out.has_camods);
	const uint16_t cprov = (uint16_t)out.carr_local;
	uint16_t Pn = NO_WSLOT;
	/* a pan ramp pending when the plan is made also gets its own line step, so that the
	 * time-parallel path sees it as one more ramp (any event recompiles the voice's plan) */
	const bool pan_ramp = (cm.pan.flags & LP_GOAL) != 0;
	if (out.has_camods || pan_ramp) { /* generator.c:756-771 */
		Pn = c.alloc();
		WideStep &pl = c.emit(ST_LINE, out.carr_local);
		pl.which = L_PAN; pl.out = Pn; pl.tmp = L_PAN; pl.flags |= SF_FORCE;
		if (out.has_camods) c.children(camods, Pn, F, F != NO_WSLOT ? cprov : NO_WSLOT, false, true);
		WideStep &v = c.emit(ST_VOICE, out.carr_local);
		v.out = V; v.pm = Pn;
	} else {
		/* no pan modulators: the carrier's own step writes the mixer row */
		c.wsteps.back().which |= OX_VOICE;
	}
	c.release(Pn); c.release(F); c.release(V);
	c.finish();
	out.n_main = c.high + 1;
	out.n_slots = out.n_main + c.high_f;
	/* closed-form (time-parallel) evaluation assumes one visit per operator */
	out.no_fast = false;
	out.static_block = false;
	out.selfmod = false;
	for (uint32_t id : out.op_ids) {
		const OpMirror &m = ops[id];
		if (m.type == SAU_POPT_N_raseg) out.static_block = true;
		if (m.type == SAU_POPT_N_noise && m.wave == SAU_NOISE_N_re) out.static_block = true;
	}
	out.ras_cub = false;
	for (uint32_t id : out.op_ids) if (ops[id].ras_cub_seen) out.ras_cub = true;
	out.n_chain = 0;
	out.n_osc = 0;
	for (const Step &st : out.steps) {
		if (out.wide) break; /* (block loop only: none of these counts is used) */
		if (step_may_chain(st)) ++out.n_chain;
		if (st.kind == ST_OSC && st.op < out.op_ids.size()) {
			const OpMirror &om = ops[out.op_ids[st.op]];
			const uint8_t ty = om.type;
			/* (red noise is a running sum too: a look-back row like an oscillator's, k_fast_voice.h) */
			if (ty == SAU_POPT_N_wave || ty == SAU_POPT_N_raseg || (ty == SAU_POPT_N_noise && om.wave == SAU_NOISE_N_re)) ++out.n_osc;
		}
		if (st.kind == ST_SMLINE) { out.static_block = true; out.selfmod = true; }
		if (st.kind == ST_LINE && st.which == L_FREQ && (st.flags & SF_FORCE)) out.static_block = true;
	}
	{
		std::vector<uint8_t> seen(out.op_ids.size(), 0);
		for (const Step &st : out.steps) {
			if (st.kind == ST_WIDE) continue;
			if (st.kind == ST_ZERO) out.no_fast = true;
			if (st.kind == ST_OSC) {
				if (seen[st.op]) out.no_fast = true;
				seen[st.op] = 1;
			}
		}
	}
	out.fast_ids.assign(out.steps.size(), FastIds());
	out.fast_ids_full.assign(out.steps.size(), FastIds());
	out.n_fast = out.n_fast_full = 0xffffffffu;
	if (!out.wide) {
		out.n_fast = fast_slot_compact(out.steps.data(), (uint32_t)out.steps.size(),
				out.steps.empty() ? nullptr : out.fast_ids.data(), false);
		/* (with frequency blocks -- the lean form: lines that are one value for every segment get none, sau_dev_types.h) */
		uint8_t ramped[256];
		for (size_t i = 0; i < 256; ++i) {
			ramped[i] = 7;
			if (i < out.op_ids.size()) {
				const OpMirror &m = ops[out.op_ids[i]];
				ramped[i] = (uint8_t)((m.freq_goal_seen ? 1 : 0) | (m.goal_seen ? 2 : 0) | (m.freq_ratio_seen ? 4 : 0));
			}
		}
		static const bool lean = !tune_env("SAU_AMD_NO_LEAN_IDS");
		if (tune_env("SAU_AMD_LEAN_IDS_LIE")) for (size_t i = 0; i < 256; ++i) ramped[i] = 0; /* (test: the device finds out, k_decode.h) */
		out.n_fast_full = fast_slot_compact(out.steps.data(), (uint32_t)out.steps.size(),
				out.steps.empty() ? nullptr : out.fast_ids_full.data(), true, lean ? ramped : nullptr);
	}
	if (out.n_fast == 0xffffffffu || out.n_fast_full == 0xffffffffu) {
		out.n_fast = out.n_fast_full = 0;
		out.no_fast = true;
	}
	return !c.failed;
}

} /* namespace sauengine */
