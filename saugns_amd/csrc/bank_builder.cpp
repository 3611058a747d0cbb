/* bank_builder.cpp -- sauProgram structures for synthetic voice banks without the parser.
 *
 * for a 1024-voice bank the reference's sau_build_Program
 * (sau/parser.c:2092) takes longer than this backend needs for the whole 10 s render, so a host
 * that generates its voices (rather than reading scripts) can describe them as a flat array of
 * operator descriptions and get the same structures the parser would have laid out for the
 * equivalent script: one event per voice, operator data in post-order with the carrier last, ids
 * in pre-order, line and time flags as sau/parser.c gives them to freshly created operators.
 * (saugns_amd/voicebank.py builds the same thing in Python; tests compare the two, and both against
 * parser-made images of BASELINE configs 2, 3 and 5.)
 */
#include "../../include/saugns_amd.h"
#include <stdlib.h>
#include <string.h>
#include <new>
#include <vector>

namespace {

struct Built { /* one allocation that owns everything the program points to */
	sauProgram prg;
	std::vector<sauProgramEvent> events;
	std::vector<sauProgramOpData> ods;
	std::vector<sauLine> lines;
	std::vector<uint32_t> idarrs;
};

sauLine make_line(const sauAmdLineDesc &d, uint32_t time_ms, bool range_partner) {
	sauLine l;
	memset(&l, 0, sizeof l);
	l.v0 = d.v0;
	l.vt = d.has_goal ? d.goal : 0.f;
	l.time_ms = time_ms;
	l.type = d.shape;
	uint8_t fl = SAU_LINEP_TIME | SAU_LINEP_TIME_IF_NEW | SAU_LINEP_STATE;
	if (!range_partner) fl |= SAU_LINEP_TYPE;
	if (d.ratio) fl |= SAU_LINEP_STATE_RATIO;
	if (d.has_goal) fl |= SAU_LINEP_GOAL;
	l.flags = fl;
	return l;
}

} /* namespace */

extern "C" sauProgram *sauAmd_build_bank(const sauAmdOpDesc *ops, size_t n_ops, float ampmult,
		uint32_t default_mod_ms) {
	if (!ops || !n_ops) return nullptr;
	/* children per operator and use type, in the order given */
	std::vector<std::vector<uint32_t>> kids(n_ops);
	std::vector<uint32_t> carriers;
	auto line_ok = [](const sauAmdLineDesc &d) { return !d.present || d.shape < SAU_LINE_NAMED; };
	for (size_t i = 0; i < n_ops; ++i) {
		const sauAmdOpDesc &o = ops[i];
		/* ids the generator would index tables with: operator type, wave / noise id, R line and function, ramp shapes */
		if (o.type >= SAU_POPT_TYPES) return nullptr;
		if (o.type == SAU_POPT_N_wave && (o.mode & 0xff) >= SAU_WAVE_NAMED) return nullptr;
		if (o.type == SAU_POPT_N_wave && o.mode > 0xff) return nullptr;
		if (o.type == SAU_POPT_N_noise && o.mode >= SAU_NOISE_NAMED) return nullptr;
		if (o.type == SAU_POPT_N_raseg && ((o.mode & 0xff) >= SAU_LINE_NAMED || ((o.mode >> 16) & 0x3f) >= SAU_RAS_FUNCTIONS ||
		                                  (o.mode >> 22) != 0)) return nullptr;
		if (!line_ok(o.pan) || !line_ok(o.amp) || !line_ok(o.amp2) || !line_ok(o.freq) || !line_ok(o.freq2) || !line_ok(o.pm_a))
			return nullptr;
		/* a voice's length is its carrier's: it has to state one (the parser gives carriers a set time too) */
		if (o.use == SAU_POP_N_carr && o.time_ms == 0) return nullptr;
		if (o.use == SAU_POP_N_carr) carriers.push_back((uint32_t)i);
		else {
			if (o.parent >= n_ops || o.parent == i || o.use >= SAU_POP_NAMED) return nullptr;
			kids[o.parent].push_back((uint32_t)i);
		}
	}
	if (carriers.empty() || carriers.size() > 0xFFFE) return nullptr;
	Built *b = new (std::nothrow) Built();
	if (!b) return nullptr;
	b->events.resize(carriers.size());
	b->ods.reserve(n_ops);
	b->lines.reserve(n_ops * 6);
	b->idarrs.reserve(n_ops * 2 + 16);
	std::vector<uint32_t> id_of(n_ops, 0xFFFFFFFFu);
	uint32_t next_id = 0, depth_max = 0;
	uint64_t dur = 0;
	/* ids in pre-order, use types ascending, as the parser numbers nested objects */
	struct Frame { uint32_t op, depth; };
	std::vector<Frame> stack;
	for (uint32_t c : carriers) {
		stack.push_back({c, 0});
		while (!stack.empty()) {
			Frame f = stack.back(); stack.pop_back();
			if (id_of[f.op] != 0xFFFFFFFFu) { delete b; return nullptr; } /* a cycle, or two parents */
			id_of[f.op] = next_id++;
			if (f.depth > depth_max) depth_max = f.depth;
			/* push children so that they pop in (use ascending, given order) */
			for (int use = SAU_POP_NAMED - 1; use >= 1; --use)
				for (size_t k = kids[f.op].size(); k-- > 0;)
					if (ops[kids[f.op][k]].use == (uint32_t)use) stack.push_back({kids[f.op][k], f.depth + 1});
		}
	}
	if (next_id != n_ops) { delete b; return nullptr; } /* operators not under any carrier */
	if (depth_max > 255) { delete b; return nullptr; } /* op_nest_depth is a uint8_t (sau/program.h:259) */
	/* operator data in post-order per voice */
	auto add_line = [&](const sauAmdLineDesc &d, uint32_t t_ms, bool rp) -> sauLine * {
		if (!d.present) return nullptr;
		b->lines.push_back(make_line(d, t_ms, rp));
		return &b->lines.back();
	};
	struct Visit { uint32_t op; bool expanded; };
	std::vector<Visit> vs;
	for (size_t v = 0; v < carriers.size(); ++v) {
		const size_t first_od = b->ods.size();
		vs.push_back({carriers[v], false});
		while (!vs.empty()) {
			Visit cur = vs.back();
			if (!cur.expanded) {
				vs.back().expanded = true;
				for (int use = SAU_POP_NAMED - 1; use >= 1; --use)
					for (size_t k = kids[cur.op].size(); k-- > 0;)
						if (ops[kids[cur.op][k]].use == (uint32_t)use) vs.push_back({kids[cur.op][k], false});
				continue;
			}
			vs.pop_back();
			const sauAmdOpDesc &o = ops[cur.op];
			sauProgramOpData od;
			memset(&od, 0, sizeof od);
			od.id = id_of[cur.op];
			od.params = SAU_POP_PARAMS;
			const bool carrier = o.use == SAU_POP_N_carr;
			const uint32_t t_ms = o.time_ms ? o.time_ms : default_mod_ms;
			od.time.v_ms = t_ms;
			od.time.flags = o.time_ms ? SAU_TIMEP_SET : (SAU_TIMEP_SET | SAU_TIMEP_DEFAULT | SAU_TIMEP_IMPLICIT);
			if (carrier) {
				sauAmdLineDesc pan = o.pan;
				if (!pan.present) { memset(&pan, 0, sizeof pan); pan.present = 1; pan.shape = SAU_LINE_N_lin; }
				od.pan = add_line(pan, t_ms, false);
			}
			od.amp = add_line(o.amp, t_ms, false);
			od.amp2 = add_line(o.amp2, t_ms, true);
			if (o.type == SAU_POPT_N_wave || o.type == SAU_POPT_N_raseg) {
				od.freq = add_line(o.freq, t_ms, false);
				od.freq2 = add_line(o.freq2, t_ms, true);
				od.pm_a = add_line(o.pm_a, t_ms, false);
			}
			od.phase = o.phase;
			od.seed = o.seed;
			od.use_type = (uint8_t)o.use;
			od.type = (uint8_t)o.type;
			if (o.type == SAU_POPT_N_raseg) {
				od.mode.ras.line = o.mode & 0xff;
				od.mode.ras.flags = ((o.mode >> 8) & 0x3f) | SAU_RAS_O_LINE_SET | SAU_RAS_O_FUNC_SET;
				od.mode.ras.func = (o.mode >> 16) & 0x3f;
			} else {
				od.mode.main = (uint8_t)o.mode;
			}
			for (int use = 1; use < SAU_POP_NAMED; ++use) {
				size_t at = b->idarrs.size();
				uint32_t n = 0;
				b->idarrs.push_back(0);
				for (uint32_t k : kids[cur.op]) if (ops[k].use == (uint32_t)use) { b->idarrs.push_back(id_of[k]); ++n; }
				if (!n) { b->idarrs.pop_back(); continue; }
				b->idarrs[at] = n;
				const sauProgramIDArr *arr = (const sauProgramIDArr *)(uintptr_t)(at + 1); /* index + 1, fixed up below */
				switch (use) {
				case SAU_POP_N_camod: od.camods = arr; break;
				case SAU_POP_N_amod: od.amods = arr; break;
				case SAU_POP_N_ramod: od.ramods = arr; break;
				case SAU_POP_N_fmod: od.fmods = arr; break;
				case SAU_POP_N_rfmod: od.rfmods = arr; break;
				case SAU_POP_N_pmod: od.pmods = arr; break;
				case SAU_POP_N_apmod: od.apmods = arr; break;
				case SAU_POP_N_fpmod: od.fpmods = arr; break;
				}
			}
			b->ods.push_back(od);
		}
		sauProgramEvent &ev = b->events[v];
		memset(&ev, 0, sizeof ev);
		const uint32_t start = ops[carriers[v]].start_ms;
		const uint32_t prev = v ? ops[carriers[v - 1]].start_ms : 0;
		if (start < prev) { delete b; return nullptr; } /* voices in time order */
		ev.wait_ms = start - prev;
		ev.vo_id = (uint16_t)v;
		ev.carr_op_id = id_of[carriers[v]];
		ev.op_data_count = (uint32_t)(b->ods.size() - first_od);
		ev.op_data = (const sauProgramOpData *)(uintptr_t)(first_od + 1);
		const uint32_t ct = ops[carriers[v]].time_ms ? ops[carriers[v]].time_ms : default_mod_ms;
		if ((uint64_t)start + ct > dur) dur = (uint64_t)start + ct;
	}
	/* the vectors no longer move: indices become pointers */
	auto fix = [&](const sauProgramIDArr *&p) { if (p) p = (const sauProgramIDArr *)&b->idarrs[(uintptr_t)p - 1]; };
	for (sauProgramOpData &od : b->ods) {
		fix(od.camods); fix(od.amods); fix(od.ramods); fix(od.fmods);
		fix(od.rfmods); fix(od.pmods); fix(od.apmods); fix(od.fpmods);
	}
	for (sauProgramEvent &ev : b->events) ev.op_data = &b->ods[(uintptr_t)ev.op_data - 1];
	memset(&b->prg, 0, sizeof b->prg);
	b->prg.events = b->events.data();
	b->prg.ev_count = b->events.size();
	b->prg.mode = SAU_PMODE_AMP_DIV_VOICES;
	b->prg.vo_count = (uint16_t)carriers.size();
	b->prg.op_count = (uint32_t)n_ops;
	b->prg.op_nest_depth = (uint8_t)(depth_max > 255 ? 255 : depth_max);
	if (dur > 0xFFFFFFFFull) { delete b; return nullptr; } /* duration_ms is 32 bits */
	b->prg.duration_ms = (uint32_t)dur;
	b->prg.ampmult = ampmult;
	b->prg.name = "bank";
	return &b->prg; /* (first member of Built) */
}

extern "C" void sauAmd_free_bank(sauProgram *prg) {
	if (prg) delete reinterpret_cast<Built *>(prg);
}
