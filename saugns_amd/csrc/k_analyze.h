/* k_analyze.h -- part of hip_backend.hip (included there, inside namespace sauhip; not a header of its own).
 * analyze_kernel: which voices the time-parallel path takes this segment, and in which form. */

/* a W oscillator step whose self-modulation is on (generator.c:479-498, wosc.h:273-310): chain_kernel's; or an R
 * oscillator's (rasg.h:242-294): rchain_kernel's -- on this path only when its inputs are its own lines
 * (step_is_chain_inline; analyze_kernel sends every other voice with R feedback to the block loop) */
__device__ __forceinline__ bool step_is_chain(const Step &st, const DevOp &o) {
	return !o.rt_frozen && step_may_chain(st) && (o.type == OT_WAVE || o.type == OT_RASEG) &&
		(st.sm != NO_SLOT || o.line[L_PMA].v0 != 0.f || (o.line[L_PMA].flags & LP_GOAL));
}
/* ... and whose varying frequency is its only phase input: chain_kernel sums the phase increments itself
 * (a sum pass and a scan less), the chain-input pass hands it increments instead of base phases */
__device__ __forceinline__ bool step_is_chain_acc(const Step &st, const DevOp &o) {
	return step_is_chain(st, o) && !o.rt_fconst_valid && st.pm == NO_SLOT && st.fpm == NO_SLOT;
}

/* ... and all of whose inputs are its own lines: a frequency that is one value, or its frequency line alone
 * (times a parent frequency that is one value) with nothing added into its block, amounts from its pm_a line.
 * chain_kernel's feeder wave evaluates those itself; the chain-input pass has nothing to do for it.
 * line_step: the plan index of the ST_LINE step that fills its frequency block, or ~0u. */
__device__ __forceinline__ bool step_is_chain_inline(bool enabled, const Step *plan, uint32_t si, const uint32_t *ids, const DevOp *ops,
		uint32_t *line_step) {
	if (!enabled) { *line_step = ~0u; return false; }
	const Step st = plan[si];
	const DevOp &o = ops[ids[st.op]];
	*line_step = ~0u;
	if (!step_is_chain(st, o) || st.pm != NO_SLOT || st.fpm != NO_SLOT || st.sm != NO_SLOT) return false;
	if (o.rt_fconst_valid) return true;
	uint32_t fmul = st.fmul, prov = st.prov;
	if (st.freq != NO_SLOT) {
		uint32_t q = si;
		bool found = false;
		while (q-- > 0) { /* its block: written by its own line step and by nothing since */
			const Step sq = plan[q];
			if (sq.kind == ST_LINE && sq.which == L_FREQ && sq.op == st.op && sq.out == st.freq) { found = true; break; }
			if ((sq.kind == ST_OSC || sq.kind == ST_LERP || sq.kind == ST_LINE || sq.kind == ST_SMLINE) && sq.out == st.freq) return false;
		}
		if (!found) return false;
		*line_step = q;
		fmul = plan[q].fmul; prov = plan[q].prov;
	}
	if (fmul != NO_SLOT) { /* a ratio of the parent's frequency: only when that is one value */
		const LineState &fl = o.line[L_FREQ];
		const bool ratio = (fl.flags & LP_STATE_RATIO) || ((fl.flags & LP_GOAL) && (fl.flags & LP_GOAL_RATIO));
		if (ratio && !(prov != NO_SLOT && ops[ids[prov]].rt_fconst_valid)) return false;
	}
	return true;
}

/* the operator whose frequency line most recently filled block `slot` before step si (0xff: none) */
__device__ __forceinline__ uint32_t block_owner(const Step *plan, uint32_t si, uint32_t slot) {
	for (uint32_t q = si; q-- > 0;) {
		const Step sq = plan[q];
		if (sq.kind == ST_LINE && sq.which == L_FREQ && sq.out == slot) return sq.op;
	}
	return 0xff;
}

/* does voice-local operator `op` take frequency-scaled phase modulation? */
__device__ __forceinline__ bool op_has_fpm(const Step *plan, uint32_t n, uint32_t op) {
	for (uint32_t q = 0; q < n; ++q) {
		const Step sq = plan[q];
		if (sq.kind == ST_OSC && sq.op == op) return sq.fpm != NO_SLOT;
	}
	return false;
}

/* Round 6: a workgroup (one wave) per voice, the voice's operator records and plan staged in LDS. The analysis is one thread's
 * chain of dependent reads and read-modify-writes of the 256-byte records -- 25 us for a config-3 segment and 53 us for a config-4
 * one when every one of them went to HBM (a thread per voice) --; the wave copies them in (64 lanes, 16 bytes
 * each), lane 0 runs the analysis on the copies, the wave copies them back. lds_ops: operator records the launch's LDS holds per
 * voice (0, or fewer than this voice has: the analysis runs on the records in HBM as before). */
__global__ void __launch_bounds__(64) analyze_kernel(FastParams P, uint32_t lds_ops, uint32_t lds_steps) {
	extern __shared__ __align__(16) unsigned char a_lds[];
	const uint32_t v = blockIdx.x;
	const int lane = threadIdx.x;
	if (v == 0 && lane == 0 && P.inmix) /* the XCDs' task queues of the closed-form launch (k_fast_types.h) */
		for (uint32_t x = 0; x < 8; ++x) P.inmix[INMIX_QUEUE + INMIX_LINE * x] = 0;
	if (v == 0 && lane == 0) { P.work_count[0] = 0; P.work_count[1] = 0; } /* finalize_kernel (a later launch) builds the block loop's work list; [1]: premix_kernel's verdict */
	if (v >= P.n_voices) return;
	const VoiceDesc vd = P.voices[v];
	const uint32_t *gids = P.op_ids + vd.ops_ofs;
	const bool use_lds = lds_ops != 0 && vd.nops <= lds_ops && vd.plan_len <= lds_steps; /* (uniform over the wave) */
	DevOp *lops = (DevOp *)a_lds;
	Step *lplan = (Step *)(a_lds + (size_t)lds_ops * sizeof(DevOp));
	uint32_t *lids = (uint32_t *)(a_lds + (size_t)lds_ops * sizeof(DevOp) + (size_t)lds_steps * sizeof(Step)); /* 0, 1, 2 ...: the copies' indices */
	/* per block buffer (256 ids): extra lead-in lanes; bit 0 / 1: depends on a chain's output / on a chain not fed from its own lines;
	 * the deepest running-sum level it depends on -- bytes in LDS (until round 6: bit planes in registers, thirty instructions a look) */
	__shared__ uint8_t a_extra[256], a_dep[256], a_level[256];
	for (uint32_t i = (uint32_t)lane; i < 64; i += 64) { ((uint32_t *)a_extra)[i] = 0; ((uint32_t *)a_dep)[i] = 0; ((uint32_t *)a_level)[i] = 0; }
	if (use_lds) {
		static_assert(sizeof(DevOp) % 16 == 0 && sizeof(Step) == 16, "copied 16 bytes at a time");
		constexpr uint32_t Q = sizeof(DevOp) / 16;
		for (uint32_t i = (uint32_t)lane; i < vd.nops * Q; i += 64)
			((uint4 *)lops)[i] = ((const uint4 *)&P.ops[gids[i / Q]])[i % Q];
		for (uint32_t i = (uint32_t)lane; i < vd.plan_len; i += 64) ((uint4 *)lplan)[i] = ((const uint4 *)(P.steps + vd.plan_ofs))[i];
		for (uint32_t i = (uint32_t)lane; i < vd.nops; i += 64) lids[i] = i;
	}
	__syncthreads();
	if (lane == 0) {
	const uint32_t *ids = use_lds ? lids : gids; /* (for the helpers that take an id list and a record array) */
	DevOp *const OPS = use_lds ? lops : P.ops;
	bool bad = (vd.flags & VD_NO_FAST) != 0 || !P.enable;
	bool seq = false, has_red = false, has_rcub = false, has_rchain = false;
	uint32_t min_time = 0xFFFFFFFFu;
	const Step *plan = use_lds ? lplan : P.steps + vd.plan_ofs;
	/* An operator that has run out of time yields nothing, and neither it nor
	 * anything nested in it advances (run_block gives its subtree zero
	 * length, generator.c:686-700): such subtrees are left out below. */
	for (uint32_t i = 0; i < vd.nops; ++i) OPS[ids[i]].rt_frozen = 0;
	if (P.chain_desc)
		for (uint32_t k = 0; k < vd.n_chain; ++k) /* ChainDesc.n (its first word; the type is defined further down) */
			((uint32_t *)P.chain_desc)[(size_t)(vd.chain_slot + k) * CHAIN_DESC_WORDS] = 0;
	const bool chain_ok = P.chain_rows != nullptr && P.scan != nullptr;
	bool has_chain = false;
	{
		uint32_t dep = 0, frozen_at = 0;
		for (uint32_t si = 0; si < vd.plan_len; ++si) {
			const Step st = plan[si];
			DevOp &o = OPS[ids[st.op]];
			if (st.flags & SF_BEGIN) {
				++dep;
				if (!frozen_at && !(o.flags & OPF_TIME_INF) && o.time == 0) frozen_at = dep;
			}
			if (frozen_at) o.rt_frozen = 1;
			if (st.flags & SF_END) {
				if (dep == frozen_at) frozen_at = 0;
				--dep;
			}
		}
	}
	if (OPS[ids[vd.carr_local]].rt_frozen) bad = true; /* the voice is over (generator.c:839) */
	for (uint32_t i = 0; i < vd.nops; ++i) {
		DevOp &o = OPS[ids[i]];
		if (o.rt_frozen) continue;
		/* ramps in progress: amplitude lines are closed-form per frame (sau/line.c
		 * fills depend on the position only); frequency ramps need a phase scan,
		 * self-modulation and pan ramps stay with the block loop */
		/* `cub` with the reference build's loop tails (sau_dev_math.h: TailCtx): which samples take the tail form depends
		 * on where the reference's blocks end; the block loop walks them, the closed forms here do not */
		/* (an R oscillator's `cub` segments are a map over whole blocks: closed-form voices with one take a build of the
		 * time-parallel kernel with that code, FastInfo.cub; a `cub` sweep in progress is a fill that ends where the
		 * sweep does: block loop) */
		if ((vd.flags & VD_TAILS) && o.type == OT_RASEG && o.wave == LN_cub) has_rcub = true;
		for (uint32_t ln = 0; ln < L_COUNT; ++ln) {
			if (!(o.line[ln].flags & LP_GOAL)) continue;
			if ((vd.flags & VD_TAILS) && o.line[ln].type == LN_cub) bad = true;
			if (ln == L_FREQ || ln == L_FREQ2) seq = true; /* phase becomes a running sum */
			else if (ln == L_PAN) { /* fine when the plan gives the pan line a step of its own */
				if (!(vd.plan_len && plan[vd.plan_len - 1].kind == ST_VOICE)) bad = true;
			} else if (ln == L_PMA) { if (!(chain_ok && (o.type == OT_WAVE || (o.type == OT_RASEG && P.chain_early_ok)))) bad = true; }
			else if (ln != L_AMP && ln != L_AMP2) bad = true;
		}
		/* red noise (noise.h:136-147) is a wrapping running sum of a counter hash: the single-pass build takes it
		 * like a running-sum phase, prefixes by look-back (has_red: only there) */
		if (o.type == OT_NOISE && o.wave == NZ_re) { seq = true; has_red = true; }
		/* self-modulation is a recurrence: W oscillators' go to chain_kernel, R's to rchain_kernel when their inputs are
		 * their own lines (decided per step below), else to the block loop */
		if (o.line[L_PMA].v0 != 0.f && !(chain_ok && (o.type == OT_WAVE || (o.type == OT_RASEG && P.chain_early_ok)))) bad = true;
		if (o.type == OT_WAVE) o.ras_level = 0; /* (CHAIN_MARK of an earlier segment) */
		o.rt_fconst_valid = 0;
		o.rt_fblk_valid = 0;
		o.st_phase = 0; /* until the kernels stage into it: see "modulated blocks" below */
		o.st_prev_phase = 0; /* likewise: extra lead-in of the operator while this kernel and decode_kernel run */
		if (!(o.flags & OPF_TIME_INF) && o.time < min_time) min_time = o.time;
	}
	uint32_t depth = 0, maxd = 0;
	/* Extra lead-in per block buffer, over what its writer's nesting depth gives: contents exact
	 * from lane H - depth + 1 + extra. 0..7 in three bit planes over the 256 buffer ids. It
	 * arises where a ratio frequency multiplies by a modulated frequency block written at a
	 * smaller depth than the reader's (see "modulated block" below) and travels up the
	 * operator tree with the data. */
	auto extra_of = [&](uint32_t sl) -> uint32_t { return sl == NO_SLOT ? 0u : (uint32_t)a_extra[sl & 255u]; };
	auto set_extra = [&](uint32_t sl, uint32_t x, bool keep_max) {
		if (sl == NO_SLOT) return;
		if (keep_max) { const uint32_t old = extra_of(sl); if (old > x) x = old; }
		a_extra[sl & 255u] = (uint8_t)(x & 7u);
	};
	uint32_t x_carrier = 0;
	for (uint32_t si = 0; si < vd.plan_len && !bad; ++si) {
		const Step st = plan[si];
		DevOp &o = OPS[ids[st.op]];
		if (o.rt_frozen) { /* nesting still counts: depths of live steps stay what they are */
			if (st.flags & SF_BEGIN) ++depth;
			if (st.flags & SF_END) --depth;
			continue;
		}
		if (st.flags & SF_BEGIN) { ++depth; if (depth > maxd) maxd = depth; }
		const bool is_osc = o.type == OT_WAVE || o.type == OT_RASEG;
		const bool freq_here = (st.kind == ST_LINE && st.which == L_FREQ) ||
			(st.kind == ST_OSC && st.freq == NO_SLOT && is_osc);
		if (st.kind == ST_ZERO) bad = true;
		if (st.kind == ST_SMLINE && !(chain_ok && o.type == OT_WAVE)) bad = true;
		if (st.kind == ST_OSC && st.sm != NO_SLOT && !(chain_ok && o.type == OT_WAVE)) bad = true;
		if (step_may_chain(st) && o.type == OT_WAVE &&
		    (st.sm != NO_SLOT || o.line[L_PMA].v0 != 0.f || (o.line[L_PMA].flags & LP_GOAL)))
			has_chain = true;
		if (step_may_chain(st) && o.type == OT_RASEG && !o.rt_frozen &&
		    (st.sm != NO_SLOT || o.line[L_PMA].v0 != 0.f || (o.line[L_PMA].flags & LP_GOAL))) {
			has_chain = true; /* (its eligibility -- fed from its own lines -- is decided with the chains' data flow below) */
			has_rchain = true;
		}
		/* a ratio line (sau/line.c:72) multiplies by the parent's frequency: one value, or a block */
		bool pconst = false; float pf = 0.f;
		if (st.fmul != NO_SLOT && st.fmul >= FSLOT_BASE) {
			/* the parent's frequency block as it stands when this step reads it: one value if
			 * the parent's line is held and nothing has been added into the block yet (the
			 * first FM modulator of a plain carrier sees exactly that, generator.c:448-477) */
			const uint32_t ow = block_owner(plan, si, st.fmul);
			if (ow != 0xff) {
				const DevOp &po = OPS[ids[ow]];
				pconst = po.rt_fblk_valid != 0; pf = po.rt_fconst;
			} else if (st.prov != NO_SLOT) {
				const DevOp &po = OPS[ids[st.prov]];
				pconst = po.rt_fconst_valid != 0; pf = po.rt_fconst;
			}
		}
		/* Anything added into a frequency block makes it per-frame: a modulated block. Its
		 * contents are exact only from the lane its writers are (nesting depth wd: lane
		 * H - wd + 1). Whoever multiplies by it -- ratio lines of operators nested in its
		 * owner, directly or through blocks derived from it -- must not need it earlier:
		 * a reader at depth d sums increments from lane H - d + 1 on (one earlier when it
		 * scales a phase modulator by its frequency). A deeper reader gets that many lanes of
		 * extra lead-in (and so does everything that consumes its output, up to the carrier:
		 * the voice's rows get H + extra lead-in lanes).
		 * (st_phase holds wd while this kernel runs; 0: not a modulated block.) */
		uint32_t x_step = 0; /* this step's output: extra lead-in of what it reads */
		if (st.kind == ST_LERP) x_step = extra_of(st.freq) > extra_of(st.pm) ? extra_of(st.freq) : extra_of(st.pm);
		if ((st.kind == ST_OSC || st.kind == ST_LERP) && st.out != NO_SLOT && st.out >= FSLOT_BASE) {
			const uint32_t ow = block_owner(plan, si, st.out);
			if (ow != 0xff) {
				DevOp &oo = OPS[ids[ow]];
				oo.rt_fblk_valid = 0;
				const uint32_t wd = st.kind == ST_OSC ? depth : depth + 1;
				if (oo.st_phase == 0 || wd < oo.st_phase) oo.st_phase = wd;
			}
		}
		if (st.kind == ST_LINE || freq_here) {
			const LineState &ls = o.line[st.kind == ST_LINE ? st.which : L_FREQ];
			const bool g_ratio = (ls.flags & LP_GOAL_RATIO) != 0, s_ratio = (ls.flags & LP_STATE_RATIO) != 0;
			if (st.fmul != NO_SLOT) {
				/* a ramp whose goal and state disagree about being ratios rescales its
				 * state by the parent's first sample (sau/line.c:358-370): block loop */
				if ((ls.flags & LP_GOAL) && g_ratio != s_ratio) {
					/* fine when the parent's frequency is one value for the segment: decode_kernel and
					 * finalize_kernel then apply the rescaling with it */
					const bool parent_const = st.prov != NO_SLOT && OPS[ids[st.prov]].rt_fconst_valid != 0;
					if (!parent_const) bad = true;
				}
				if ((s_ratio || ((ls.flags & LP_GOAL) && g_ratio)) && !pconst) {
					seq = true;
					if (st.fmul >= FSLOT_BASE) {
						const uint32_t ow = block_owner(plan, si, st.fmul);
						const uint32_t wd = ow != 0xff ? OPS[ids[ow]].st_phase : 0u;
						if (wd) {
							const uint32_t need = depth + (op_has_fpm(plan, vd.plan_len, st.op) ? 1u : 0u);
							/* the block is exact from lane H - wd + 1 + its own extra; this reader
							 * would sum from lane H - need + 1 */
							x_step = (need > wd ? need - wd : 0u) + extra_of(st.fmul);
							/* this operator's own block derives from the modulated one (its second frequency's line too: the
							 * range modulation mixes it into the block, generator.c:466-467) */
							if (st.kind == ST_LINE && (st.which == L_FREQ || st.which == L_FREQ2) && (o.st_phase == 0 || wd < o.st_phase))
								o.st_phase = wd;
						}
					}
				}
			}
		}
		if (freq_here && !bad) {
			const LineState &fl = o.line[L_FREQ];
			/* one value for the segment? (the block loop's const_freq) */
			bool isconst = !(fl.flags & LP_GOAL) && !(st.kind == ST_LINE && (st.flags & SF_FORCE));
			float fc = fl.v0;
			if (st.fmul != NO_SLOT && (fl.flags & LP_STATE_RATIO)) {
				if (pconst) fc = fl.v0 * pf; /* sau/line.c:72 */
				else isconst = false;
			}
			o.rt_fconst = fc;
			o.rt_fconst_valid = isconst ? 1u : 0u;
			/* the block itself (before modulators are added) holds one value? */
			o.rt_fblk_valid = (isconst || (st.kind == ST_LINE && (st.flags & SF_FORCE) && !(fl.flags & LP_GOAL) &&
					!((fl.flags & LP_STATE_RATIO) && st.fmul != NO_SLOT && !pconst))) ? 1u : 0u;
			if (!isconst) seq = true;
		}
		if (st.kind == ST_OSC && is_osc && st.freq != NO_SLOT && !o.rt_fconst_valid) seq = true;
		/* extra lead-in: what this step reads, what its own ratio frequency needs, to what it writes */
		if (st.kind == ST_LINE) {
			/* the block made here is as exact as the one it multiplies by; what the operator itself
			 * needs on top waits in st_prev_phase for its oscillator step */
			if (x_step > 7) bad = true;
			set_extra(st.out, extra_of(st.fmul), false);
			/* (the second frequency's line as well -- round 4: a ratio f2 under a modulated parent frequency, on an oscillator
			 * with frequency-scaled PM, dropped its lane here; the oscillator's value on its first defined lane was then
			 * wrong in every row, which shows when a repeated phase on the next lane copies it: batch 3883 of
			 * tests/tools/gpu_vs_ref_batches.py, one frame of a pan modulator, only under some segment cuts) */
			if ((st.which == L_FREQ || st.which == L_FREQ2) && x_step > o.st_prev_phase) o.st_prev_phase = x_step;
		} else if (st.kind == ST_LERP) {
			if (x_step > 7) bad = true;
			set_extra(st.out, x_step, true);
		} else if (st.kind == ST_OSC) {
			uint32_t x = x_step > o.st_prev_phase ? x_step : o.st_prev_phase; /* its own frequency's need */
			const uint32_t in[5] = {extra_of(st.freq), extra_of(st.pm), extra_of(st.fpm), extra_of(st.amp), extra_of(st.sm)};
#pragma unroll
			for (int k = 0; k < 5; ++k) if (in[k] > x) x = in[k];
			if (x > 7) bad = true;
			o.st_prev_phase = x; /* for decode_kernel (the kernels stage into this field only later) */
			if (st.op == vd.carr_local) x_carrier = x;
			if (!(st.which & OX_VOICE)) set_extra(st.out, x, (st.flags & SF_LAYER) != 0 || st.out >= FSLOT_BASE);
		} else if (st.kind == ST_VOICE) {
			const uint32_t x = extra_of(st.out) > extra_of(st.pm) ? extra_of(st.out) : extra_of(st.pm);
			if (x > x_carrier) x_carrier = x;
		}
		if (st.flags & SF_END) --depth;
	}
	/* Two passes suffice when no running sum depends on another one: the per-frame
	 * increments of every such oscillator (its frequency inputs) must not depend on
	 * the output of an oscillator whose phase is itself a running sum. Forward
	 * data-flow over the block buffers ("tainted" = depends on such an output). */
	/* Several passes instead of one wave in order: a running sum can be computed by all
	 * waves once the sums it depends on are known. Level 1: its per-frame increments (its
	 * frequency inputs) depend on no other running-sum oscillator's output; level n + 1:
	 * they depend on level-n outputs. Forward data-flow over the block buffers, two bits
	 * per buffer: the deepest level its contents depend on. */
	/* Feedback chains: the recurrence's inputs (frequency, phase modulators, amounts) must not depend on any
	 * chain's output, and no running sum may either -- the sum passes and the chain-input pass run before
	 * chain_kernel. Forward data-flow, one bit per block buffer ("depends on a chain's output"). */
	bool early = false;
	if (has_chain && !bad) {
		/* c0: depends on some chain's output; c1: ... on the output of a chain that is not fed from its own lines.
		 * Chains of the latter kind run in chunks between the chain-input and final passes, after the sum passes:
		 * nothing those passes compute may depend on them. The former kind can run before everything else (early). */
		constexpr uint32_t c0 = 1u, c1 = 2u; /* (bits of a_dep) */
		auto dep = [&](uint32_t c, uint32_t sl) -> bool { return sl != NO_SLOT && (a_dep[sl & 255u] & c) != 0; };
		auto set_dep = [&](uint32_t c, uint32_t sl, bool v, bool keep) {
			if (sl == NO_SLOT) return;
			const uint8_t old = a_dep[sl & 255u];
			a_dep[sl & 255u] = v ? (uint8_t)(old | c) : (keep ? old : (uint8_t)(old & ~c));
		};
		for (uint32_t si = 0; si < vd.plan_len && !bad; ++si) {
			const Step st = plan[si];
			const DevOp &o = OPS[ids[st.op]];
			if (o.rt_frozen) continue;
			if (st.kind == ST_LINE) { set_dep(c0, st.out, dep(c0, st.fmul), false); set_dep(c1, st.out, dep(c1, st.fmul), false); }
			else if (st.kind == ST_SMLINE) { set_dep(c0, st.out, false, false); set_dep(c1, st.out, false, false); }
			else if (st.kind == ST_LERP) {
				set_dep(c0, st.out, dep(c0, st.freq) || dep(c0, st.pm), true);
				set_dep(c1, st.out, dep(c1, st.freq) || dep(c1, st.pm), true);
			} else if (st.kind == ST_OSC) {
				const bool in_dep = dep(c0, st.pm) || dep(c0, st.fpm) || dep(c0, st.freq) || dep(c0, st.fmul) || dep(c0, st.sm);
				const bool in_late = dep(c1, st.pm) || dep(c1, st.fpm) || dep(c1, st.freq) || dep(c1, st.fmul) || dep(c1, st.sm);
				const bool chain = step_may_chain(st) && (o.type == OT_WAVE || o.type == OT_RASEG) &&
					(st.sm != NO_SLOT || o.line[L_PMA].v0 != 0.f || (o.line[L_PMA].flags & LP_GOAL));
				uint32_t ls_ = ~0u;
				const bool inl = chain && step_is_chain_inline(P.chain_early_ok != 0, plan, si, ids, OPS, &ls_);
				/* (a chain that sums its own increments is no running sum of the passes) */
				const bool fvar = (o.type == OT_WAVE || o.type == OT_RASEG) && !o.rt_fconst_valid && !(chain && inl);
				if (chain && o.type == OT_RASEG) { /* R feedback: only as an early chain fed from its own lines */
					if (inl) early = true; else bad = true;
				}
				if (chain && in_late) bad = true;
				else if (chain && in_dep) early = true;
				if (fvar && (dep(c1, st.freq) || dep(c1, st.fmul))) bad = true;
				else if (fvar && (dep(c0, st.freq) || dep(c0, st.fmul))) early = true;
				if (!(st.which & OX_VOICE)) {
					set_dep(c0, st.out, chain || in_dep || dep(c0, st.amp), (st.flags & SF_LAYER) != 0);
					set_dep(c1, st.out, (chain && !inl) || in_late || dep(c1, st.amp), (st.flags & SF_LAYER) != 0);
				}
			}
		}
	}
	uint32_t seq_kind = seq ? 1u : 0u, n_scan_out = 0, levels_out = 0, lvl_bits_out = 0;
	if (seq && !bad) {
		auto level_of = [&](uint32_t sl) -> uint32_t { return sl == NO_SLOT ? 0u : (uint32_t)a_level[sl & 255u]; };
		auto set_level = [&](uint32_t sl, uint32_t lv, bool keep_max) {
			if (sl == NO_SLOT) return;
			if (keep_max) { const uint32_t old = level_of(sl); if (old > lv) lv = old; }
			a_level[sl & 255u] = (uint8_t)(lv & 3u);
		};
		auto max2 = [](uint32_t a, uint32_t b) { return a > b ? a : b; };
		bool multi = true;
		uint32_t n_scan = 0;
		for (uint32_t si = 0; si < vd.plan_len; ++si) {
			const Step st = plan[si];
			DevOp &o = OPS[ids[st.op]];
			if (o.rt_frozen) continue;
			if (st.kind == ST_LINE) {
				set_level(st.out, level_of(st.fmul), false);
			} else if (st.kind == ST_LERP) {
				set_level(st.out, max2(level_of(st.freq), level_of(st.pm)), true);
			} else if (st.kind == ST_OSC) {
				const bool fvar = ((o.type == OT_WAVE || o.type == OT_RASEG) && !o.rt_fconst_valid &&
					!(chain_ok && step_is_chain_acc(st, o))) || (o.type == OT_NOISE && o.wave == NZ_re);
				uint32_t lv = max2(max2(level_of(st.pm), level_of(st.fpm)), max2(level_of(st.amp),
						max2(level_of(st.freq), level_of(st.fmul))));
				if (fvar) {
					const uint32_t mine = 1 + max2(level_of(st.freq), level_of(st.fmul));
					if (mine > P.sum_levels || n_scan >= FAST_MAX_SCAN) multi = false; /* deeper: one wave, in order */
					else {
						lvl_bits_out |= mine << (2 * n_scan);
						if (mine > levels_out) levels_out = mine;
						o.rt_fblk_valid = mine; /* (its first meaning is over: from here on the operator's level) */
					}
					++n_scan;
					lv = max2(lv, mine > 3 ? 3u : mine);
				}
				if (!(st.which & OX_VOICE)) set_level(st.out, lv, (st.flags & SF_LAYER) != 0);
			}
		}
		if (P.look && !has_chain && n_scan <= FAST_MAX_SCAN && n_scan <= vd.n_look) {
			seq_kind = 3; /* one pass, any depth: prefixes by look-back */
		} else if (multi && P.scan) {
			seq_kind = 2;
			for (uint32_t p = 0; p < levels_out && p < P.sum_levels; ++p) atomicOr(&P.pass_flags[p], 1u);
		}
		n_scan_out = n_scan;
	}
	if (has_red && seq_kind != 3) bad = true; /* (the several-pass and in-order forms do not carry noise sums) */
	(void)has_rchain;
	if (has_rcub && (!P.cub_ok || (seq_kind != 0 && seq_kind != 3) || has_chain)) bad = true; /* (only the closed-form build has the tail code) */
	if (has_chain && !bad) {
		if (seq_kind == 1) bad = true; /* (one wave in order: not with chains) */
		else {
			seq_kind = 2; /* multi-pass voice, possibly without sums */
			atomicOr(&P.pass_flags[FAST_MAX_LEVELS + 1], 1u);
		}
	}
	FastInfo fi;
	/* Running-sum voices get one more lead-in lane than their data flow needs: a repeated phase on
	 * the first lane an operator is defined in then spoils nothing that is stored (what it spoils
	 * climbs one lane per nesting level and ends on the lane before the first stored one), where
	 * closed-form voices have repair_kernel for that case (see FAST_REPAIR_SHIFT). */
	if (seq || has_chain) ++x_carrier;
	fi.n_chain = has_chain && !bad ? 1u : 0u;
	fi.H = maxd + x_carrier; fi.xlead = x_carrier; fi.bail = 0; fi.n_fsteps = 0; fi.n_pass[0] = fi.n_pass[1] = fi.n_pass[2] = fi.n_pass[3] = 0; fi.seq = seq_kind; fi.cub = has_rcub ? 1u : 0u; fi.early = (early && has_chain && !bad) ? 1u : 0u; fi.n_scan = n_scan_out; fi.levels = levels_out; fi.lvl_bits = lvl_bits_out; fi.tail = 0; fi.tail_stream = 0;
	fi.total = 0;
	if ((seq || has_chain) && !P.seq_enable) bad = true;
	if (!bad && vd.nops <= P.max_ops && vd.plan_len <= P.max_steps && maxd >= 1 && maxd + x_carrier <= P.np / 2)
		fi.total = min(min_time, vd.run_len);
	P.info[v] = fi;
	if (fi.total && seq_kind == 2 && n_scan_out == 0 && P.lean_on) atomicOr(&P.pass_flags[FAST_LEAN_FLAG], 1u);
	else if (fi.total && (seq_kind == 1 || seq_kind == 2)) atomicOr(&P.pass_flags[FAST_MAX_LEVELS + 2], 1u);
	P.fast_done[v] = 0;
	P.repair[(size_t)v * FAST_REPAIR_WORDS] = 0;
	if (fi.total && fi.cub) atomicOr(&P.pass_flags[FAST_CUB_FLAG], 1u);
	if (fi.total && fi.early) atomicOr(&P.pass_flags[FAST_EARLY_FLAG], 1u);
	if (P.split_cf && fi.total) { /* the two launches' voice lists (any order: voices are independent) */
		if (seq_kind == 0) P.vlists[atomicAdd(&P.pass_flags[FAST_CF_COUNT], 1u)] = v;
		else if (seq_kind == 3) P.vlists[P.n_voices + atomicAdd(&P.pass_flags[FAST_LK_COUNT], 1u)] = v;
	}
	} /* (lane 0) */
	if (use_lds) { /* the records back, with what the analysis has noted in them for decode_kernel and the launches */
		__syncthreads();
		constexpr uint32_t Q = sizeof(DevOp) / 16;
		for (uint32_t i = (uint32_t)lane; i < vd.nops * Q; i += 64)
			((uint4 *)&P.ops[gids[i / Q]])[i % Q] = ((const uint4 *)lops)[i];
	}
}

