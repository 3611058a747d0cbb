/* sau_dev_types.h -- data shared between the host control plane, the HIP
 * kernels and the test-side sequential executor: device operator state, event
 * update records, and the flattened per-voice plan ("batched operator graph").
 */
#ifndef SAU_DEV_TYPES_H
#define SAU_DEV_TYPES_H

#include "sau_dev_math.h"

namespace saudev {

/* Line slots inside an operator, same order as the sweep ids of the program
 * format (sau/program.h:53-60). */
enum : uint32_t { L_PAN = 0, L_AMP, L_AMP2, L_FREQ, L_FREQ2, L_PMA, L_COUNT };

enum : uint32_t { /* DevOp.flags */
	OPF_TIME_INF = 1u << 0,  /* generator.c:42 ON_TIME_INF */
	OPF_OSC_RESET = 1u << 1, /* wosc.h:37-38 */
	OPF_RATE2X = 1u << 2,    /* rasg.h:32 */
};

/* Persistent state of one operator (generator.c:45-88 OperatorNode, flattened).
 * Lives in HBM between launches; a voice's operators are cached in LDS while
 * its workgroup renders a segment. 64 dwords. */
struct DevOp {
	uint32_t time = 0;       /* remaining samples (unless OPF_TIME_INF) */
	uint32_t flags = 0;      /* OPF_* */
	uint32_t type = 0;       /* OT_* */
	uint32_t wave = 0;       /* W: wave id; N: noise id; R: line shape */
	LineState line[L_COUNT] = {}; /* 36 dwords */
	float coeff = 0;         /* 2^32 / srate as f32 (wosc.h:30, rasg.h:27) */
	uint32_t phase = 0;      /* W: accumulator (includes the wave's phase_adj) */
	uint32_t prev_phase = 0; /* W */
	uint32_t ras_flags = 0;  /* R */
	double prev_Is = 0;      /* W */
	uint64_t cycle_phase = 0;/* R */
	float prev_s = 0, fb_s = 0; /* W and R feedback */
	uint32_t ras_func = 0, ras_level = 0, ras_alpha = 0;
	uint32_t noise_n = 0, noise_prev = 0; /* N */
	/* per-block scratch of the renderer (not part of the persistent state):
	 * set while this operator's frequency is one value for the whole block */
	float rt_fconst = 0;
	uint32_t rt_fconst_valid = 0;
	/* end-of-segment oscillator state staged by the time-parallel path */
	uint32_t st_prev_phase = 0;
	double st_prev_Is = 0;
	float st_prev_s = 0;
	/* per-segment: this operator or one it is nested in has run out of time, so
	 * it produces nothing and its state stands still (generator.c:686-700) */
	uint32_t rt_frozen = 0;
	uint32_t st_phase = 0;   /* accumulator at the segment's last frame (sequential scan) */
	uint32_t rt_fblk_valid = 0; /* per-segment: its frequency block holds one value (rt_fconst) so far */
};
static_assert(sizeof(DevOp) == 256, "DevOp is 64 dwords");

/* One operator update of an event (sauProgramOpData by value), applied on the
 * device by the event kernel: generator.c:245-343 prepare_op + update_op. */
struct OpUpdate {
	uint32_t op;
	uint32_t params;       /* SAU_POPP_* */
	uint32_t type;         /* OT_* */
	uint32_t first;        /* operator not yet initialised: run prepare_op */
	uint32_t time;         /* samples, when params & TIME */
	uint32_t time_inf;
	uint32_t mode_main;    /* wave / noise id */
	uint32_t ras_line, ras_flags, ras_func, ras_level, ras_alpha;
	uint32_t phase, seed;
	float coeff;
	uint32_t loop_tails;   /* the reference build's loop tails are reproduced (sau_dev_math.h: TailCtx); was padding */
	LineUpdate line[L_COUNT];
};

enum : uint32_t { POPP_TIME = 1, POPP_MODE = 2, POPP_PHASE = 4, POPP_SEED = 8 };

/* Per-wave constants, sau/wave.h:33-69,144-149 */
struct WaveConst {
	float diff_scale;  /* amp_scale * 0.125f * (float)UINT32_MAX */
	float diff_offset; /* amp_dc */
	int32_t phase_adj;
	uint32_t pad;
};

/* ---- plan ------------------------------------------------------------------ */

constexpr uint8_t NO_SLOT = 0xFF;
constexpr uint8_t SCRATCH_SLOT = 0; /* used inside a step only */
constexpr uint8_t FSLOT_BASE = 128; /* slot ids from here: frequency blocks */

/* slot id -> index of its memory, for a launch with n_main main-pool slots */
SAU_HD uint32_t slot_index(uint32_t id, uint32_t n_main) {
	return id < FSLOT_BASE ? id : n_main + (id - FSLOT_BASE);
}

enum : uint8_t {
	ST_LINE = 1, /* out <- line `which` of op (x fmul)               */
	ST_LERP,     /* out += (freq - out) * pm   (generator.c:466-467)  */
	ST_OSC,      /* evaluate operator, combine into out               */
	ST_ZERO,     /* out <- 0 (circular reference guard, gen.c:685-689)*/
	ST_SMLINE,   /* out <- pm_a line or zeros; latch self-mod state   */
	ST_VOICE,    /* hand carrier block (+pan) to the mixer            */
	ST_WIDE,     /* high bytes of the previous step's ids (wide plans)  */
};

enum : uint8_t {
	SF_BEGIN = 1 << 0,     /* first step of op: clip len to op time, push  */
	SF_END = 1 << 1,       /* last step of op: zero tail, time -= len, pop  */
	SF_WAVE_ENV = 1 << 2,  /* block_mix_mul_waveenv instead of _add          */
	SF_LAYER = 1 << 3,     /* combine with existing out instead of replacing */
	SF_SKIP2 = 1 << 4,     /* ST_LINE: also skip the range partner line      */
	SF_FORCE = 1 << 3,     /* ST_LINE: always write the slot (it is added to) */
	SF_SKIP_FREQ2 = 1 << 4,/* ST_OSC with inline freq: skip freq2            */
	SF_SKIP_AMP2 = 1 << 5, /* ST_OSC with inline amp: skip amp2              */
	SF_SM_INLINE = 1 << 6, /* ST_OSC: test/run pm_a line inline (no apmods)  */
	SF_SM_SKIP = 1 << 7,   /* ST_OSC: pm_a line was set once: skip it        */
};

enum : uint8_t { /* Step.which for ST_OSC */
	OX_VOICE = 1 << 0,     /* carrier: hand the block to the mixer in this step */
};

struct Step {
	uint8_t kind, flags;
	uint8_t out;   /* destination slot */
	uint8_t freq;  /* ST_OSC: freq slot or NO_SLOT (inline); ST_LERP: range end */
	uint8_t fmul;  /* parent frequency slot for ratio lines, or NO_SLOT */
	uint8_t pm;    /* ST_OSC: PM sum; ST_LERP: modulator; ST_VOICE: pan slot */
	uint8_t fpm;
	uint8_t amp;   /* ST_OSC: amp slot or NO_SLOT (inline) */
	uint8_t sm;    /* ST_OSC: self-modulation amount slot */
	uint8_t which; /* ST_LINE: L_* */
	uint8_t tmp;   /* extra scratch slot (R self-mod) */
	uint8_t prov;  /* voice-local op whose frequency block `fmul` is, or NO_SLOT */
	uint32_t op;   /* voice-local operator index */
};
static_assert(sizeof(Step) == 16, "Step is 4 dwords");

/* Block buffers as the time-parallel path sees them. It reads a step's inputs
 * into registers before it stores the step's output, so far fewer buffers are
 * live at once than the block loop's plan names: a depth-D modulator chain
 * needs one. This walks a plan, frees each buffer at its last read and hands
 * out the lowest free number. Two flavours: without frequency blocks (every
 * oscillator frequency is one value for the segment: they are never
 * materialised) and with them (frequency ramps, FM: the sequential-scan mode).
 * Returns the count of compact buffers (0xffffffff: more than 64); when `ids`
 * is given, ids[i] receives step i's buffers (NO_SLOT where unused). Runs on
 * the host when a plan is compiled; the ids travel to the device beside the
 * steps. */
struct FastIds { uint8_t out, pm, fpm, amp, aux /* LERP range end */, freq, fmul, sm /* self-modulation amounts */; };
static_assert(sizeof(FastIds) == 8, "FastIds is 2 dwords");
SAU_HD uint32_t fast_slot_compact(const Step *plan, uint32_t n, FastIds *ids, bool with_freq, const uint8_t *ramped = nullptr) {
	const uint32_t limit = with_freq ? 250u : (uint32_t)FSLOT_BASE;
	constexpr int NR = 7;
	uint8_t last[256];
	uint8_t map[256];
	/* `ramped` (with_freq; indexed by voice-local operator: its frequency line has had a ramp at some time): the lean form --
	 * a frequency line that is one value for every segment gets no buffer, as decode_kernel will not materialise it
	 * (k_decode.h: keep; analyze_kernel: rt_fconst_valid). One value: never ramped, nothing added into its block
	 * (SF_FORCE: an FM or range modulator will be), and -- a ratio line multiplies by the parent's block -- the block it
	 * may multiply by one value too when this step reads it (analyze_kernel: block_owner, rt_fblk_valid). one[s]: plan
	 * buffer s holds such a frequency at this point of the plan (1: not materialised, 2: a real block nothing has been
	 * added into yet). The carrier-FM bank needs 2 buffers for 4, BASELINE config 4's second voice 4 for 5: rows per pass
	 * are what LDS holds of them. Should the device find such a line not to be one value after all, the voice goes to the
	 * block loop (decode_kernel). */
	uint8_t one[256];
	const bool lean = with_freq && ramped != nullptr;
	for (uint32_t s = 0; s < 256; ++s) { last[s] = 0xff; map[s] = NO_SLOT; one[s] = 0; }
	auto reads_of = [](const Step &st, uint8_t *rd) {
		for (int k = 0; k < NR; ++k) rd[k] = NO_SLOT;
		if (st.kind == ST_OSC) { rd[0] = st.pm; rd[1] = st.fpm; rd[2] = st.amp; if (st.flags & SF_LAYER) rd[3] = st.out; rd[4] = st.freq; rd[5] = st.fmul; rd[6] = st.sm; }
		else if (st.kind == ST_LERP) { rd[0] = st.out; rd[1] = st.freq; rd[2] = st.pm; }
		else if (st.kind == ST_VOICE) { rd[0] = st.out; rd[1] = st.pm; }
		else if (st.kind == ST_LINE) { rd[0] = st.fmul; }
	};
	/* (lean: which frequency lines get no buffer, decided in plan order ahead of the liveness pass -- what reads such a line's
	 * buffer reads nothing) */
	uint8_t skip[256], strict[256];
	for (uint32_t i = 0; i < 256; ++i) { skip[i] = 0; strict[i] = 0; }
	if (lean) {
		for (uint32_t i = 0; i < n && i < 0xff; ++i) {
			const Step st = plan[i];
			const bool writes_blk = (st.kind == ST_OSC && !(st.which & OX_VOICE)) || st.kind == ST_LERP;
			const uint8_t rf = st.op < 256 ? ramped[st.op] : (uint8_t)7; /* (256 entries; 1: a frequency line has ramped, 2: some line has, 4: a frequency line has been a ratio) */
			if (st.kind == ST_LINE && st.which == L_FREQ && st.out < limit) {
				const bool parent_one = st.fmul == NO_SLOT || st.fmul >= limit || !(rf & 4) || one[st.fmul] != 0;
				const bool own = !(rf & 1);
				if (own && parent_one && !(st.flags & SF_FORCE)) { one[st.out] = 1; skip[i] = 1; }
				else one[st.out] = (own && parent_one) ? 2 : 0;
				/* (... and by the narrower rule decode_kernel can check again when a range end multiplies by this block: its own
				 * multiplier, if it has one, a line without a buffer) */
				strict[st.out] = own && (st.fmul == NO_SLOT || st.fmul >= limit || !(rf & 4) || one[st.fmul] == 1);
			} else if (st.kind == ST_LINE && (st.which == L_FREQ2 || st.which == L_AMP2) && st.out < limit && !(rf & 2) && /* (SF_FORCE on these: "always written" -- for the blend's sake) */
			           (st.fmul == NO_SLOT || st.fmul >= limit || !(rf & 4) || one[st.fmul] == 1 || (one[st.fmul] == 2 && strict[st.fmul]))) {
				/* the end of a range (generator.c:466-467) that is one value: folded into the blend that reads it, if nothing
				 * else does before the buffer is written again (decode_kernel: ST_LERP with the end as a constant) */
				bool only_blends = false;
				for (uint32_t j = i + 1; j < n && j < 0xff; ++j) {
					const Step sj = plan[j];
					uint8_t rj[NR];
					reads_of(sj, rj);
					bool other = false;
					for (int k = 0; k < NR; ++k) if (rj[k] == st.out && !(sj.kind == ST_LERP && k == 1)) other = true;
					if (other) { only_blends = false; break; }
					if (sj.kind == ST_LERP && sj.freq == st.out) only_blends = true;
					const bool wj = (sj.kind == ST_OSC && !(sj.which & OX_VOICE)) || sj.kind == ST_LERP || sj.kind == ST_LINE || sj.kind == ST_SMLINE;
					if (wj && sj.out == st.out) break;
				}
				if (only_blends) skip[i] = 1;
				one[st.out] = 0;
			} else if ((writes_blk || st.kind == ST_LINE || st.kind == ST_SMLINE) && st.out < limit) {
				one[st.out] = 0; /* something else is written or added into it */
			}
		}
	}
	/* (a buffer skipped this way is never live: its readers are not counted) */
	uint8_t dead[256]; /* per plan buffer, as the walk below goes: its current value is a skipped line's */
	for (uint32_t s = 0; s < 256; ++s) dead[s] = 0;
	for (uint32_t i = 0; i < n && i < 0xff; ++i) {
		uint8_t rd[NR];
		reads_of(plan[i], rd);
		if (skip[i]) { dead[plan[i].out] = 1; continue; }
		const Step st = plan[i];
		const bool writes_new = ((st.kind == ST_OSC && !(st.which & OX_VOICE) && !(st.flags & SF_LAYER)) || st.kind == ST_LINE || st.kind == ST_SMLINE);
		for (int k = 0; k < NR; ++k) if (rd[k] < limit && !dead[rd[k]]) last[rd[k]] = (uint8_t)i;
		if (writes_new && st.out < limit) dead[st.out] = 0;
	}
	for (uint32_t s = 0; s < 256; ++s) dead[s] = 0;
	unsigned long long used = 0;
	uint32_t count = 0;
	if (ids) {
		FastIds none;
		none.out = none.pm = none.fpm = none.amp = none.aux = none.freq = none.fmul = none.sm = NO_SLOT;
		for (uint32_t i = 0; i < n; ++i) ids[i] = none;
	}
	for (uint32_t i = 0; i < n && i < 0xff; ++i) {
		const Step st = plan[i];
		if (!with_freq && st.kind == ST_LINE && st.which == L_FREQ) continue; /* never materialised */
		if (skip[i]) { map[st.out] = NO_SLOT; dead[st.out] = 1; continue; } /* (lean: one value, never materialised either) */
		uint8_t rd[NR];
		reads_of(st, rd);
		for (int k = 0; k < NR; ++k) if (rd[k] < limit && dead[rd[k]]) rd[k] = NO_SLOT; /* (reads of a skipped line's buffer) */
		bool writes = false;
		if (st.kind == ST_OSC) writes = !(st.which & OX_VOICE);
		else if (st.kind == ST_LERP || st.kind == ST_LINE || st.kind == ST_SMLINE) writes = true;
		FastIds got;
		got.out = got.pm = got.fpm = got.amp = got.aux = got.freq = got.fmul = got.sm = NO_SLOT;
		if (st.kind == ST_OSC) {
			if (st.pm < limit) got.pm = map[st.pm];
			if (st.fpm < limit) got.fpm = map[st.fpm];
			if (st.amp < limit) got.amp = map[st.amp];
			if (st.freq < limit) got.freq = map[st.freq];
			if (st.fmul < limit) got.fmul = map[st.fmul];
			if (st.sm < limit) got.sm = map[st.sm];
		} else if (st.kind == ST_LERP) {
			if (st.freq < limit) got.aux = map[st.freq];
			if (st.pm < limit) got.pm = map[st.pm];
		} else if (st.kind == ST_VOICE) {
			if (st.pm < limit) got.pm = map[st.pm];
		} else if (st.kind == ST_LINE) {
			if (st.fmul < limit) got.fmul = map[st.fmul];
		}
		/* inputs read for the last time here give their buffer back before the
		 * output is placed (the output may then land on one of them) */
		for (int k = 0; k < NR; ++k) {
			const uint8_t s = rd[k];
			if (s < limit && last[s] == i && map[s] != NO_SLOT && !(writes && s == st.out))
				used &= ~(1ull << map[s]); /* map[s] stays: a later write to s is a new value */
		}
		if ((writes || st.kind == ST_VOICE) && st.out < limit) {
			if (writes && (map[st.out] == NO_SLOT || !((used >> map[st.out]) & 1ull))) {
				uint32_t c = 0;
				while (c < 64 && ((used >> c) & 1ull)) ++c;
				if (c >= 64) return 0xffffffffu;
				used |= 1ull << c;
				map[st.out] = (uint8_t)c;
				if (c + 1 > count) count = c + 1;
				if (last[st.out] == 0xff || last[st.out] < i) used &= ~(1ull << c); /* nobody reads it */
			}
			got.out = map[st.out];
			if (writes) dead[st.out] = 0;
		}
		if (ids) ids[i] = got;
	}
	return count;
}

/* Wide plans. A step names its buffers in 8 bits: 127 main and 122 frequency ids, which a straight modulator
 * chain of about 120 levels uses up (every level holds its frequency block and its modulator sum while the
 * levels below run). The reference takes 256 levels (sauProgram.op_nest_depth is a uint8: sau/program.h:259,
 * generator.c:133 gives each level 7 buffers). A voice whose graph needs more ids gets a *wide* plan: every
 * step is followed by an ST_WIDE step that holds the high bytes of the ids (and of `prov`) in the same
 * fields; 16-bit ids, main pool 1..0x7FFE, frequency pool from WSLOT_FBASE, NO_WSLOT = none. Only the block
 * loop runs wide plans (VD_WIDE voices are VD_NO_FAST too); it reads every step through step_widen(), ids
 * turned into memory indices on the way. */
constexpr uint16_t NO_WSLOT = 0xFFFF;
constexpr uint16_t WSLOT_FBASE = 0x8000;
struct WideStep {
	uint8_t kind, flags, which;
	uint16_t out, freq, fmul, pm, fpm, amp, sm;
	uint16_t tmp;  /* ST_OSC: scratch buffer (memory index); lines: the partner line */
	uint16_t prov; /* voice-local operator or NO_WSLOT */
	uint32_t op;
};
SAU_HD uint16_t wslot_index(uint32_t lo, uint32_t hi, bool wide, uint32_t n_main) {
	if (!wide) return lo == NO_SLOT ? NO_WSLOT : (uint16_t)slot_index(lo, n_main);
	const uint32_t id = lo | (hi << 8);
	if (id == NO_WSLOT) return NO_WSLOT;
	return (uint16_t)(id < WSLOT_FBASE ? id : n_main + (id - WSLOT_FBASE));
}
/* `hi`: the ST_WIDE step behind `lo` in a wide plan, else null */
SAU_HD WideStep step_widen(const Step &lo, const Step *hi, uint32_t n_main) {
	const bool w = hi != nullptr;
	WideStep s;
	s.kind = lo.kind; s.flags = lo.flags; s.which = lo.which; s.op = lo.op;
	s.out = wslot_index(lo.out, w ? hi->out : 0, w, n_main);
	s.freq = wslot_index(lo.freq, w ? hi->freq : 0, w, n_main);
	s.fmul = wslot_index(lo.fmul, w ? hi->fmul : 0, w, n_main);
	s.pm = wslot_index(lo.pm, w ? hi->pm : 0, w, n_main);
	s.fpm = wslot_index(lo.fpm, w ? hi->fpm : 0, w, n_main);
	s.amp = wslot_index(lo.amp, w ? hi->amp : 0, w, n_main);
	s.sm = wslot_index(lo.sm, w ? hi->sm : 0, w, n_main);
	if (lo.kind == ST_OSC) s.tmp = wslot_index(lo.tmp, w ? hi->tmp : 0, w, n_main);
	else s.tmp = lo.tmp;
	if (w) { const uint32_t p = lo.prov | ((uint32_t)hi->prov << 8); s.prov = (uint16_t)p; }
	else s.prov = lo.prov == NO_SLOT ? NO_WSLOT : lo.prov;
	return s;
}

/* A self-modulated W oscillator ("chain": wosc.h:273-310) as the time-parallel path handles it: its
 * inputs (base phases, self-modulation amounts) go to a pair of rows in HBM, chain_kernel runs the
 * per-sample recurrence with one lane per chain, the final pass reads the samples back. */
SAU_HD bool step_may_chain(const Step &st) {
	return st.kind == ST_OSC && ((st.flags & SF_SM_INLINE) || st.sm != NO_SLOT);
}
/* One voice of one render stream, for one segment launch. */
struct VoiceDesc {
	uint32_t plan_ofs, plan_len; /* into the step array */
	uint32_t ops_ofs, nops;      /* into the op-id array (global op indices) */
	uint32_t carr_local;         /* carrier's voice-local op index */
	uint32_t run_len;            /* min(duration, segment length) */
	uint32_t out_row;            /* row in the voice output matrix */
	uint32_t pan_dynamic_row;    /* row of the pan matrix, or ~0u */
	uint32_t flags;              /* VD_* */
	Lattice lat;                 /* where the reference's blocks lie in this segment, for the voice's program */
	uint32_t chain_base, n_chain;/* row pairs for its self-modulated oscillators (step_may_chain steps, in plan order) */
	uint32_t chain_slot;         /* ... and their lanes of the chain kernels' waves (ChainDesc entries): the engine begins every kind of R
	                              * feedback on a wave of its own, so the lanes are numbered with gaps and the rows without */
	uint32_t inc_base, n_inc;    /* row pairs for saved phase increments of its oscillator steps (in plan order), or n_inc = 0 */
	uint32_t look_base, n_look;  /* look-back rows for its running-sum oscillators (a voice without feedback chains), or n_look = 0 */
	uint32_t ev_left;            /* frames from the segment's first frame to its program's next event (~0u: none): a reference
	                              * block ends there whichever span it lies in (TailCtx) */
};

enum : uint32_t {
	VD_NO_FAST = 1u << 0, /* graph visits an operator twice or has a cycle guard */
	VD_MORE = 1u << 1,    /* the voice goes on after this segment (its blocks are not cut at run_len: TailCtx.rem) */
	VD_TAILS = 1u << 2,   /* the reference build's loop tails are reproduced */
	VD_WIDE = 1u << 3,    /* wide plan: step pairs with 16-bit buffer ids (step_widen) */
};

/* Per-voice result of a segment, read by the mixer. */
struct VoiceOut {
	float pan_const;    /* pan.v0 when no pan row is written */
	uint32_t has_pan;   /* 1: pan matrix row holds per-sample values */
	uint32_t valid_len; /* frames of the row that were written (rest is silence) */
	uint32_t pan_row;   /* row of the pan matrix when has_pan */
};

constexpr uint32_t MAX_NEST = 256; /* operators on a path from the carrier down: sauProgram.op_nest_depth, the deepest level's number, is a uint8 (sau/program.h:259) */

} /* namespace saudev */
#endif
