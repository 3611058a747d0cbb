/* k_fast_group.h -- part of hip_backend.hip: the evaluation of one row group in fast_voice (k_fast_voice.h), included there
 * once or twice with the compile-time constant EDGE in scope: true = the group may touch an end of the segment (the code as it
 * always was: in-segment tests, state carried in at the first group and staged at the last); false = it does not, and all of
 * that is compiled out (SPLIT builds: see k_fast_voice.h). Plain text inclusion, not a lambda: as a generic lambda the two
 * spilling builds fast_kernel<5, 3> and <6, 3> ran into hipcc's "requires even aligned vector registers" error on a reloaded
 * 64-bit spill (ROCm 7.2), although they take one copy only. In scope: everything fast_voice has defined up to its row-group
 * loop, plus cg, repair_rows, cgm. */
		auto in_seg = [&](int t) -> bool { return !EDGE || (t >= 0 && t < (int)fast_total); }; /* frame t lies in the segment */
		auto in_end = [&](int t) -> bool { return !EDGE || t < (int)fast_total; };              /* ... not behind its end */
		const int t0g = (int)(cg * GF) - (int)H + l - (REPAIR ? (int)FAST_REPAIR_SHIFT : 0); /* this lane's frame in row 0 */
		/* does this lane's frame of row k belong to the group (lead-in lanes: rows that have them, i.e. row 0 when CONTIG)? */
		auto own = [&](int k) -> bool { return (CONTIG && k > 0) || l >= (int)H; };
		/* REPAIR: a frame this pass is to store -- one of the noted ones among the group's first owned frames, which lie
		 * FAST_REPAIR_SHIFT lanes further on in this evaluation's row 0 */
		auto repair_mine = [&](int k) -> bool {
			const int j = l - (int)(H + FAST_REPAIR_SHIFT);
			return k == 0 && j >= 0 && j < 32 && ((repair_rows >> (j & 31)) & 1u);
		};
		/* running sums: what a row's inclusive scan holds at its last lead-in lane (rows without lead-in lanes: nothing) */
		auto lead32 = [&](uint32_t Sk, int k) -> uint32_t { return (CONTIG && k > 0) ? 0u : (uint32_t)__builtin_amdgcn_readlane((int)Sk, (int)H - 1); };
		auto lead64 = [&](unsigned long long Sk, int k) -> unsigned long long { return (CONTIG && k > 0) ? 0ull : readlane64(Sk, (int)H - 1); };
		/* what the lane before holds of row k's x -- for lane 0 of a later row of a contiguous group: the row before's lane 63 */
		/* (the DPP move's lane 0 has no lane to read from: it keeps the `old` operand, here the row before's lane 63) */
		auto prev_of = [&](uint32_t cur, uint32_t before, int k) -> uint32_t {
			if (CONTIG && k > 0) { /* (two DPP moves: the row before rotated by a lane puts its lane 63 into lane 0) */
				/* (bound_ctrl for the rotation: every lane has a source, and with it the compiler needs no `old` value -- without,
				 * it zeroed the destination first, one more VALU instruction per rotated value) */
				const int rot = __builtin_amdgcn_update_dpp(0, (int)before, 0x13c /* wave_ror:1 */, 0xf, 0xf, true);
				return (uint32_t)__builtin_amdgcn_update_dpp(rot, (int)cur, 0x138 /* wave_shr:1 */, 0xf, 0xf, false);
			}
			return lane_prev(cur);
		};
		auto prev32 = [&](const uint32_t *x, int k) -> uint32_t { return prev_of(x[k], x[k > 0 ? k - 1 : 0], k); };
		auto prev64 = [&](const double *x, int k) -> double {
			const double b = x[k > 0 ? k - 1 : 0];
			const uint32_t lo = prev_of((uint32_t)__double2loint(x[k]), (uint32_t)__double2loint(b), k);
			const uint32_t hi = prev_of((uint32_t)__double2hiint(x[k]), (uint32_t)__double2hiint(b), k);
			return __hiloint2double((int)hi, (int)lo);
		};
		const bool first_group = EDGE && (cg == 0);
		const bool is_last_group = EDGE && (cg == last_group);
		uint32_t held_rows = 0; /* rows with a hold this evaluation could not resolve; the closed-form builds: which of the group's
		                         * first owned frames such holds spoil (bit j: the frame at row 0's lane H + j) */
		bool held_far = false;  /* ... or something the repair pass cannot put right (a later row, a frame further on) */
		(void)repair_rows;
#if FK_PREFETCH
		FastStep fnext = load_step_uniform(fsteps);
#endif
		for (uint32_t si = 0; si < n_fsteps; ++si) {
			const int t0 = t0g;
#if FK_PREFETCH
			const FastStep f = fnext;
			fnext = load_step_uniform(fsteps + (si + 1 < n_fsteps ? si + 1 : si)); /* in flight during this step */
#else
			const FastStep f = load_step_uniform(fsteps + si);
#endif
			const uint32_t kind = f.kind & 0xff;
			const uint32_t flags = (f.kind >> 8) & 0xff;
			const bool sum_pass = FULL && P.mode != 0 && P.mode <= P.sum_levels;
			if (sum_pass && !(f.ramp & (2u << P.mode))) continue; /* not needed for this pass's phase increments */
			const bool chain_in = CH && P.mode == P.sum_levels + 2; /* the pass that writes the chains' inputs */
			if (chain_in && !(f.ramp & FR_CHAIN_IN)) continue;
			if (CH && P.mode == P.sum_levels + 1 && (f.ramp & FR_FINAL_SKIP)) continue;
			if (kind == ST_OSC) {
				const uint32_t type = f.type & 0xff;
				const bool wave_env = (flags & SF_WAVE_ENV) != 0;
				const bool layer = (flags & SF_LAYER) != 0;
				const bool to_voice = ((f.kind >> 16) & OX_VOICE) != 0;
				float s[T];
				const bool chain = CH && (f.type & FT_CHAIN) != 0;
				if ((type == OT_WAVE && chain && (P.mode == P.sum_levels + 1 || (f.type & FT_CHAIN_EARLY))) ||
				    (type == OT_RASEG && chain)) { /* (R feedback: always an early chain, rchain_kernel) */ /* (the final pass; any pass) */
					/* a feedback chain: chain_kernel has run it; its samples are in the row */
					const float *crow = P.chain_rows + (size_t)2 * f_bits(f.pan) * P.chain_stride;
#pragma unroll
					for (int k = 0; k < T; ++k) {
						const int t = t0 + k * (int)RS;
						s[k] = in_seg(t) ? FK_CLOAD(&crow[t]) : 0.f;
					}
				} else if (type == OT_WAVE) {
					const bool has_pm = f.pm_off != ~0u, has_fpm = f.fpm_off != ~0u;
					/* this operator's values are defined from lane p_min on
					 * (one more lead-in sample per nesting level below it) */
					const int p_min = (int)H - (int)(f.kind >> 24) + 1;
					bool done = false;
					if (FK_COMMON && f.tab >= 0 && !has_fpm && !first_group && !is_last_group && !(f.ramp & 2) && !chain) {
						/* the common case, straight-line: table in LDS, plain PM or
						 * none, no segment edge in this group */
						uint32_t ph[T];
						{
							uint32_t acc = f.phase0 + f.inc * (uint32_t)(t0 + 1);
							const uint32_t row_inc = f.inc * RS;
#pragma unroll
							for (int k = 0; k < T; ++k) { ph[k] = acc; acc += row_inc; }
						}
						bool ok = true;
						if (has_pm) {
							float pm[T];
							bool big = false;
#pragma unroll
							for (int k = 0; k < T; ++k) {
								pm[k] = slots[f.pm_off + k * 64];
								big |= !(fabsf(pm[k]) < 0x1p20f);
							}
							ok = !__any(big);
#pragma unroll
							for (int k = 0; k < T; ++k) ph[k] += rint32w_p31_small(pm[k]);
						}
						if (ok) {
							const uint32_t ltab = tabs + (uint32_t)f.tab * FkTab<WIDE>::BYTES;
							double Is[T];
#pragma unroll
							for (int k = 0; k < T; ++k) Is[k] = fk_poly(fk_entry<WIDE>(ltab, ph[k]), ph[k]);
							if (FK_CONSTD && !has_pm && f.inc != 0) {
								/* unmodulated: every phase step is inc, one division serves all */
								const double x = (double)div_f32_normal(f.diff_scale, (float)(int32_t)f.inc);
#pragma unroll
								for (int k = 0; k < T; ++k)
									s[k] = (float)((Is[k] - prev64(Is, k)) * x + (double)f.diff_offset);
								done = true;
							} else {
								/* (round 4: the unsigned minimum of the steps -- v_min3_u32, one compare per group instead of one per
								 * row -- rendered wrong samples from the second row group on, with the DPP move folded into the
								 * subtraction; the per-row compare stays) */
								bool zero = false, zero_n = false; /* a phase step of zero in row 0 / in a later row */
								/* (round 5: one v_min_u32 per later row -- a 2-cycle instruction, tools/valu_probe.hip -- and one compare per
								 * group, where a v_cmp + s_or per row stood: 1.787 -> 1.755 ms per config-3 launch. The unsigned minimum
								 * of the steps is zero exactly when one of them is. Round 4's v_min3 form rendered wrong samples with the
								 * DPP move folded into the subtraction that fed it: the empty asm keeps the move a move) */
								uint32_t dmin = 0xffffffffu;
#pragma unroll
								for (int k = 0; k < T; ++k) {
									uint32_t pp = prev32(ph, k);
									asm("" : "+v"(pp));
									const int32_t d = (int32_t)(ph[k] - pp);
									if (CONTIG && k > 0) dmin = min(dmin, (uint32_t)d); else zero |= (d == 0);
									s[k] = wosc_diff(Is[k], prev64(Is, k), d, f.diff_scale, f.diff_offset);
								}
								zero_n = dmin == 0;
								/* (every lane of a contiguous group's later rows is a defined one) */
								done = !__any((zero && l >= p_min) || zero_n);
							}
						}
					}
					if (!done) {
						uint32_t ph[T];
						double Is[T];
						float fv[T]; /* frequency per frame (freq-scaled PM reads it) */
						bool fvar = false;
						if (SCAN && (f.ramp & 2)) {
							/* the frequency varies (ramp, FM): phase is a running sum of per-frame
							 * increments (wosc.h:135-169). This wave walks the voice's rows in order;
							 * `carry` holds the accumulator at the frame before each row's new frames. */
							const FastAux fa = load_aux_uniform(faux + si);
							fvar = (fa.flags & (FA_FVAR_SLOT | FA_FVAR_LINE)) != 0;
							if (fvar) {
								uint32_t S[T];
								auto freq_at = [&](int k, int t) -> float { /* the frequency at row k's frame t */
									if (fa.flags & FA_FVAR_SLOT) return slots[fa.freq_off + k * 64];
									float v = fast_line_value(fa.fl, t);
									const bool in_goal = (uint32_t)t < fa.fl.goal_len;
									if (fa.flags & (in_goal ? FA_MUL_GOAL : FA_MUL_HOLD))
										v *= fa.fmul_off != ~0u ? slots[fa.fmul_off + k * 64] : fa.mulc;
									return v;
								};
								/* saved increments (FastParams.inc_rows): written by the sum pass of this oscillator's
								 * level, read back by the final pass in place of the frequency */
								uint32_t *irow = (FULL && (fa.pad[2] & 2u)) ? P.inc_rows + (size_t)2 * (fa.pad[2] >> 8) * P.inc_stride : nullptr;
								const bool inc_read = irow && P.mode == P.sum_levels + 1;
								const bool inc_write = irow && two && P.mode == fa.pad[1];
								if (FULL) {
#pragma unroll
									for (int k = 0; k < T; ++k) {
										const int t = t0 + k * (int)RS;
										uint32_t r;
										if (inc_read) {
											r = in_seg(t) ? irow[t] : 0u;
											fv[k] = 0.f; /* (only frequency-scaled PM reads it, and such oscillators save nothing) */
										} else {
											const float v = freq_at(k, t);
											fv[k] = v;
											const float x = fa.coeff * v;
											/* llrintf(x) mod 2^32 (wosc.h:145): adding 1.5 * 2^52 in f64 rounds to the nearest
											 * integer and leaves it in the low word; exact while |x| < 2^51 */
											r = fabsf(x) < 0x1p50f ? (uint32_t)__double2loint((double)x + 0x1.8p52) : rint32w(x);
											if (inc_write && l >= (int)H && in_seg(t)) irow[t] = r;
										}
										const uint32_t inc = in_seg(t) ? r : 0u;
										if (chain && fa.pad[2]) S[k] = inc; /* chain_kernel does the summing */
										else S[k] = wave_incl_scan_dpp(inc);
									}
								} else { /* the single-pass build: one test per group for the rounding form */
									float x[T];
									bool big = false;
#pragma unroll
									for (int k = 0; k < T; ++k) {
										fv[k] = freq_at(k, t0 + k * (int)RS);
										x[k] = fa.coeff * fv[k];
										big |= !(fabsf(x[k]) < 0x1p50f);
									}
									uint32_t r[T];
									if (!__any(big)) {
#pragma unroll
										for (int k = 0; k < T; ++k) r[k] = (uint32_t)__double2loint((double)x[k] + 0x1.8p52);
									} else {
#pragma unroll
										for (int k = 0; k < T; ++k) r[k] = rint32w(x[k]);
									}
#pragma unroll
									for (int k = 0; k < T; ++k) {
										const int t = t0 + k * (int)RS;
										const uint32_t inc = in_seg(t) ? r[k] : 0u;
										if (SCAN == 3 && chain && fa.pad[2]) S[k] = inc; /* chain_kernel does the summing */
										else S[k] = wave_incl_scan_dpp(inc);
									}
								}
								if (chain && fa.pad[2]) {
									/* chain-input pass of a chain that accumulates its own phase: increments and amounts */
									float *brow = P.chain_rows + (size_t)2 * f_bits(f.pan) * P.chain_stride;
									float *arow = brow + P.chain_stride;
									FastLine pl;
									const bool from_line = f.aux_off == ~0u;
									if (from_line) pl = load_line_uniform(fplines + si);
#pragma unroll
									for (int k = 0; k < T; ++k) {
										const int t = t0 + k * (int)RS;
										const float a = from_line ? fast_line_value(pl, t) : slots[f.aux_off + k * 64];
										if (l >= (int)H && in_seg(t)) {
											FK_CSTORE(&((u32_alias *)brow)[t], S[k]);
											FK_CSTORE(&arow[t], a);
										}
									}
									continue;
								}
								/* the accumulator at the frame before this group's first new frame: carried by
								 * this wave (in-order voices), or the prefix of all earlier groups' sums */
								unsigned long long *sums = two ? scan + (size_t)fa.pad[0] * P.scan_groups : nullptr;
								const bool sum_me = two && P.mode == fa.pad[1]; /* this pass computes this oscillator's sums */
								uint32_t acc;
								if (look_own) {
									acc = first_group ? f.phase0 : (uint32_t)carry[si];
								} else if (look) {
									uint32_t tot = 0;
#pragma unroll
									for (int k = 0; k < T; ++k)
										tot += (uint32_t)__builtin_amdgcn_readlane((int)S[k], 63) - lead32(S[k], k);
									acc = f.phase0 + (look_lds ? lookback32<true>(lk_base + (size_t)fa.pad[0] * 2 * 64, cg, tot, 0, lk_ring, cgm, l, zero_acc)
									                           : lookback32<false>(lookv + (size_t)fa.pad[0] * 2 * P.scan_groups, cg, tot, P.look_epoch, 0, 0, l, zero_acc, (P.look_wpv_flags & 2u) != 0));
								} else {
									acc = two ? (sum_me ? 0u : f.phase0 + (uint32_t)sums[cg])
									          : (first_group ? f.phase0 : (uint32_t)carry[si]);
								}
#pragma unroll
								for (int k = 0; k < T; ++k) {
									const uint32_t lead = lead32(S[k], k);
									const uint32_t last = (uint32_t)__builtin_amdgcn_readlane((int)S[k], 63);
									ph[k] = acc + (S[k] - lead);
									acc += last - lead;
								}
								if (two) {
									if (sum_me) { /* this pass ends here for this oscillator */
										if (l == 0) sums[cg] = (unsigned long long)acc;
										continue;
									}
								} else if ((!look || look_own) && l == 0) {
									carry[si] = (unsigned long long)acc;
								}
							}
						}
						if (!fvar) {
							/* phase0 + inc*(t+1): one multiply per lane, then adds */
							uint32_t acc = f.phase0 + f.inc * (uint32_t)(t0 + 1);
							const uint32_t row_inc = f.inc * RS;
#pragma unroll
							for (int k = 0; k < T; ++k) { ph[k] = acc; acc += row_inc; fv[k] = f.fc; }
						}
						/* the accumulator after the segment's last frame, before modulation: the single-pass build stages it here;
						 * the full build keeps the values and stages them with the rest (its register allocation fares better) */
						uint32_t phu[FULL ? T : 1];
						if (FULL) {
#pragma unroll
							for (int k = 0; k < T; ++k) phu[FULL ? k : 0] = ph[k];
						} else if (SCAN && is_last_group) {
#pragma unroll
							for (int k = 0; k < T; ++k) {
								const int t = t0 + k * (int)RS;
								if (t == (int)fast_total - 1 && own(k)) P.ops[f.gop].st_phase = ph[k];
							}
						}
						if (has_pm && !has_fpm) {
							float pm[T];
							bool big = false;
#pragma unroll
							for (int k = 0; k < T; ++k) {
								pm[k] = slots[f.pm_off + k * 64];
								big |= !(fabsf(pm[k]) < 0x1p20f);
							}
							if (!__any(big)) {
#pragma unroll
								for (int k = 0; k < T; ++k) ph[k] += rint32w_p31_small(pm[k]);
							} else {
#pragma unroll
								for (int k = 0; k < T; ++k) ph[k] += rint32w_p31(pm[k]);
							}
						} else if (has_pm || has_fpm) {
							float pm[T], fpm[T];
#pragma unroll
							for (int k = 0; k < T; ++k) { pm[k] = 0.f; fpm[k] = 0.f; }
							if (has_pm) {
#pragma unroll
								for (int k = 0; k < T; ++k) pm[k] = slots[f.pm_off + k * 64];
							}
							if (has_fpm) {
#pragma unroll
								for (int k = 0; k < T; ++k) fpm[k] = slots[f.fpm_off + k * 64];
							}
							if (has_pm) {
#pragma unroll
								for (int k = 0; k < T; ++k) ph[k] += pm_offset32(true, true, pm[k], fpm[k], fv[k]);
							} else {
#pragma unroll
								for (int k = 0; k < T; ++k) ph[k] += pm_offset32(false, true, 0.f, fpm[k], fv[k]);
							}
						}
						if (chain) {
							/* chain-input pass: base phases (accumulator + phase modulation; the feedback term is
							 * chain_kernel's) and self-modulation amounts to the chain's rows, nothing else */
							float *brow = P.chain_rows + (size_t)2 * f_bits(f.pan) * P.chain_stride;
							float *arow = brow + P.chain_stride;
							FastLine pl;
							const bool from_line = f.aux_off == ~0u;
							if (from_line) pl = load_line_uniform(fplines + si);
#pragma unroll
							for (int k = 0; k < T; ++k) {
								const int t = t0 + k * (int)RS;
								const float a = from_line ? fast_line_value(pl, t) : slots[f.aux_off + k * 64];
								if (l >= (int)H && in_seg(t)) {
									FK_CSTORE(&((u32_alias *)brow)[t], ph[k]);
									FK_CSTORE(&arow[t], a);
									if (EDGE && FULL && t == (int)fast_total - 1) P.ops[f.gop].st_phase = phu[FULL ? k : 0];
								}
							}
							continue;
						}
						const bool reset = (f.type >> 16) & 1;
						if (first_group) {
							/* t = -1: the sample before the segment (wosc.h:215-231 on restart) */
							const uint32_t nxt = __shfl_down(ph[0], 1);
							if (l == (int)H - 1) ph[0] = reset ? nxt - SLEN : f.prev_phase;
						}
						if (f.tab >= 0) {
							const uint32_t ltab = tabs + (uint32_t)f.tab * FkTab<WIDE>::BYTES;
#pragma unroll
							for (int k = 0; k < T; ++k) Is[k] = fk_poly(fk_entry<WIDE>(ltab, ph[k]), ph[k]);
						} else {
							const uint32_t wave = (f.type >> 8) & 0xff;
							const HerpC23 *g23 = P.g_c23 + (size_t)wave * WAVE_LEN;
							const HerpC01 *g01 = P.g_c01 + (size_t)wave * WAVE_LEN;
#pragma unroll
							for (int k = 0; k < T; ++k) {
								const uint32_t ind = ph[k] >> SLEN_BITS;
								Is[k] = herp_poly(g23[ind], g01[ind], ph[k]);
							}
						}
						if (first_group && !reset) {
							if (l == (int)H - 1) Is[0] = f.prev_Is;
						}
						uint32_t pph[T];
						bool zero = false;
						if (FK_CONSTD && !has_pm && !has_fpm && !first_group && f.inc != 0 && !fvar) {
							/* unmodulated: every phase step is inc, one division serves all */
							const double x = (double)div_f32_normal(f.diff_scale, (float)(int32_t)f.inc);
#pragma unroll
							for (int k = 0; k < T; ++k) {
								pph[k] = ph[k] - f.inc;
								const double pIs = prev64(Is, k);
								s[k] = (float)((Is[k] - pIs) * x + (double)f.diff_offset);
							}
						} else {
#pragma unroll
							for (int k = 0; k < T; ++k) {
								pph[k] = prev32(ph, k);
								const double pIs = prev64(Is, k);
								const int32_t d = (int32_t)(ph[k] - pph[k]);
								zero |= (d == 0) && ((CONTIG && k > 0) || l >= p_min);
								s[k] = wosc_diff(Is[k], pIs, d, f.diff_scale, f.diff_offset);
							}
						}
						if (first_group && reset) {
							/* a (re)started oscillator's first sample as the reference build computes it (sau_dev_math.h:
							 * wosc_reset_s): lane H - 1 holds the phase one table step back -- its Hermite value taken apart */
							const uint32_t indp = ph[0] >> SLEN_BITS;
							FkHerp ep;
							if (f.tab >= 0) {
								ep = fk_entry<WIDE>(tabs + (uint32_t)f.tab * FkTab<WIDE>::BYTES, ph[0]);
							} else {
								const uint32_t wave = (f.type >> 8) & 0xff;
								const HerpC23 hp = (P.g_c23 + (size_t)wave * WAVE_LEN)[indp];
								const HerpC01 lp = (P.g_c01 + (size_t)wave * WAVE_LEN)[indp];
								ep.c3 = hp.c3; ep.c2 = hp.c2; ep.c1 = (double)lp.c1; ep.c0 = (double)lp.c0;
							}
							const double rise_p = lane_prev(fk_poly_rise(ep, ph[0]));
							const float c0_p = bits_f(lane_prev(f_bits((float)ep.c0))); /* (the table value: an f32, exactly) */
							if (l == (int)H) s[0] = wosc_reset_s(Is[0], rise_p, c0_p, f.diff_scale, f.diff_offset);
						}
						if (__any(zero)) {
							/* dphase == 0: the differentiator holds its previous output
							 * (wosc.h:251-252). Isolated cases resolve inside the row; a
							 * run that reaches back past the lead-in goes to the block loop. */
							bool held[T], src[T]; /* src: holds a defined output to copy from */
#pragma unroll
							for (int k = 0; k < T; ++k) {
								const int t = t0 + k * (int)RS;
								const bool defined = ((CONTIG && k > 0) || l >= p_min) && (!EDGE || t >= 0);
								held[k] = (ph[k] == pph[k]) && defined && in_end(t);
								src[k] = defined && !held[k];
							}
							for (int it = 0; it < 64; ++it) {
								bool changed = false;
#pragma unroll
								for (int k = 0; k < T; ++k) {
									float sp = __shfl_up(s[k], 1);
									bool okp = __shfl_up(src[k], 1);
									if (CONTIG && k > 0) { /* lane 0 looks at the row before's lane 63 */
										const float se = __shfl(s[k - 1], 63);
										const bool oke = __shfl(src[k - 1], 63);
										if (l == 0) { sp = se; okp = oke; }
									}
									if (held[k] && okp && (l > 0 || (CONTIG && k > 0))) { s[k] = sp; held[k] = false; src[k] = true; changed = true; }
								}
								if (!__any(changed)) break;
							}
#pragma unroll
							for (int k = 0; k < T; ++k) {
								/* running-sum voices have a lane of slack (analyze_kernel): a hold left on the
								 * operator's first defined lane is harmless there */
								if (SCAN && l == p_min && !(CONTIG && k > 0)) held[k] = false;
								if (held[k]) rep[1] = ((uint32_t)(l - p_min) << 24) | ((uint32_t)k << 20) | (si << 12) | (cg & 0xfff); /* debug */
								if (!SCAN && CONTIG) {
									/* What is left unresolved in row 0 is a run of repeats that begins on the operator's first defined lane,
									 * p_min; the hold at lane p_min + j spoils the carrier's frame at lane H + j (one lane further per nesting
									 * level on the way out). The repair pass stores exactly those frames, FAST_REPAIR_SHIFT lanes further on
									 * in its evaluation -- so they must fit a row there. (Until round 4 only the run's first frame was noted:
									 * a run of two or three at a group's start kept its later frames' garbage -- one extreme program in 3000
									 * once contiguous rows had made such groups few enough per voice to be repaired rather than redone.) */
									if (k == 0) {
										const int jm = min(32, 64 - (int)FAST_REPAIR_SHIFT - (int)H);
										const unsigned long long run = __ballot(held[0]) >> p_min;
										if (jm <= 0) { if (run) held_far = true; }
										else {
											held_rows |= (uint32_t)(run & ((1ull << jm) - 1ull));
											if (run >> jm) held_far = true;
										}
									} else if (__any(held[k])) held_far = true; /* (a later row's: the run reaches back through the rows before) */
								} else
								held_rows |= __any(held[k]) ? (1u << k) : 0u;
							}
						}
						if (is_last_group) {
							/* the row that holds the segment's last frame stages the state */
#pragma unroll
							for (int k = 0; k < T; ++k) {
								const int t = t0 + k * (int)RS;
								if (t == (int)fast_total - 1 && own(k)) {
									DevOp &o = P.ops[f.gop];
									if (FULL) o.st_phase = phu[FULL ? k : 0];
									o.st_prev_phase = ph[k];
									o.st_prev_Is = Is[k];
									o.st_prev_s = s[k];
								}
							}
						}
					}
				} else if (type == OT_RASEG) {
					/* rasg.h:165-222 + 692-743: frame t reads the counter cp0 + inc * t (+ PM) */
					const bool rate2x = (f.type >> 17) & 1;
					const float phase_scale = rate2x ? 0x1p31f * 2 : 0x1p31f;
					const RasParams rp = ras_params((uint32_t)f.tab & 0xff, ((uint32_t)f.tab >> 8) & 0xffff,
							f_bits(f.diff_scale), f_bits(f.diff_offset), ((uint32_t)f.tab >> 24) & 0x7f);
					const unsigned long long inc64 = ((unsigned long long)f.prev_phase << 32) | f.inc;
					const unsigned long long cp0 = (unsigned long long)__double_as_longlong(f.prev_Is);
					const bool has_pm = f.pm_off != ~0u, has_fpm = f.fpm_off != ~0u;
					unsigned long long cpv[T]; /* the counter each frame reads (post-increment), before PM */
					float fv[T];
					bool fvar = false;
					if (SCAN && (f.ramp & 2)) {
						/* the frequency varies: the counter is a running sum of 64-bit increments */
						const FastAux fa = load_aux_uniform(faux + si);
						fvar = (fa.flags & (FA_FVAR_SLOT | FA_FVAR_LINE)) != 0;
						if (fvar) {
							const float rcoeff = rate2x ? fa.coeff * 2 : fa.coeff;
							unsigned long long S[T], incv[T];
							uint32_t *irow = (FULL && (fa.pad[2] & 2u)) ? P.inc_rows + (size_t)2 * (fa.pad[2] >> 8) * P.inc_stride : nullptr;
							const bool inc_read = irow && P.mode == P.sum_levels + 1;
							const bool inc_write = irow && two && P.mode == fa.pad[1];
#pragma unroll
							for (int k = 0; k < T; ++k) {
								const int t = t0 + k * (int)RS;
								const bool in_sg = in_seg(t);
								if (inc_read) { /* saved by the sum pass of its level: low and high words */
									incv[k] = in_sg ? ((unsigned long long)irow[P.inc_stride + t] << 32) | irow[t] : 0ull;
									fv[k] = 0.f;
								} else {
								float v;
								if (fa.flags & FA_FVAR_SLOT) {
									v = slots[fa.freq_off + k * 64];
								} else {
									v = fast_line_value(fa.fl, t);
									const bool in_goal = (uint32_t)t < fa.fl.goal_len;
									if (fa.flags & (in_goal ? FA_MUL_GOAL : FA_MUL_HOLD))
										v *= fa.fmul_off != ~0u ? slots[fa.fmul_off + k * 64] : fa.mulc;
								}
								fv[k] = v;
								incv[k] = in_sg ? (unsigned long long)rint64(rcoeff * v) : 0ull;
								if (inc_write && l >= (int)H && in_sg) { irow[t] = (uint32_t)incv[k]; irow[P.inc_stride + t] = (uint32_t)(incv[k] >> 32); }
								}
								S[k] = wave_incl_scan64_dpp(incv[k]);
							}
							unsigned long long *sums = two ? scan + (size_t)fa.pad[0] * P.scan_groups : nullptr;
							const bool sum_me = two && P.mode == fa.pad[1];
							unsigned long long acc;
							if (look_own) {
								acc = first_group ? cp0 : carry[si];
							} else if (look) {
								unsigned long long tot = 0;
#pragma unroll
								for (int k = 0; k < T; ++k) tot += readlane64(S[k], 63) - lead64(S[k], k);
								if (look_lds) {
									unsigned long long *e_lo = lk_base + (size_t)fa.pad[0] * 2 * 64;
									acc = cp0 + lookback64<true>(e_lo, e_lo + 64, cg, tot, 0, lk_ring, cgm, l, zero_acc);
								} else {
									unsigned long long *e_lo = lookv + (size_t)fa.pad[0] * 2 * P.scan_groups;
									acc = cp0 + lookback64<false>(e_lo, e_lo + P.scan_groups, cg, tot, P.look_epoch, 0, 0, l, zero_acc, (P.look_wpv_flags & 2u) != 0);
								}
							} else {
								acc = two ? (sum_me ? 0ull : cp0 + sums[cg])
								          : (first_group ? cp0 : carry[si]);
							}
#pragma unroll
							for (int k = 0; k < T; ++k) {
								const unsigned long long lead = lead64(S[k], k);
								const unsigned long long last = readlane64(S[k], 63);
								cpv[k] = acc + (S[k] - lead) - incv[k];
								acc += last - lead;
							}
							if (two) {
								if (sum_me) {
									if (l == 0) sums[cg] = acc;
									continue;
								}
							} else if ((!look || look_own) && l == 0) {
								carry[si] = acc;
							}
							if (is_last_group) { /* the counter after the segment's last frame */
#pragma unroll
								for (int k = 0; k < T; ++k) {
									const int t = t0 + k * (int)RS;
									if (t == (int)fast_total - 1 && own(k))
										P.ops[f.gop].st_prev_Is = __longlong_as_double((long long)(cpv[k] + incv[k]));
								}
							}
						}
					}
					if (!fvar) {
#pragma unroll
						for (int k = 0; k < T; ++k) {
							const int t = t0 + k * (int)RS;
							cpv[k] = cp0 + inc64 * (unsigned long long)(long long)t;
							fv[k] = f.fc;
						}
					}
#pragma unroll
					for (int k = 0; k < T; ++k) {
						unsigned long long cp = cpv[k];
						if (has_pm || has_fpm)
							cp += (unsigned long long)pm_offset(has_pm, has_fpm,
									has_pm ? slots[f.pm_off + k * 64] : 0.f,
									has_fpm ? slots[f.fpm_off + k * 64] : 0.f, fv[k], phase_scale);
						uint32_t cyc;
						float phf;
						ras_split(cp, cyc, phf);
						bool ctail = false;
						if constexpr (CUB) {
							if (f.type & FT_CUBTAIL) { /* the last len % 4 samples of the reference's block (sau_dev_math.h: TailCtx) */
								TailCtx tc;
								tc.lat = vd.lat; tc.ev_left = vd.ev_left; tc.off = 0; tc.rem = f.phase0; tc.on = 1;
								const int t = t0 + k * (int)RS;
								ctail = t >= 0 && cub_map_is_tail(tc, (uint32_t)t);
							}
						}
						s[k] = ras_sample(rp, cyc, phf, true, ctail);
					}
				} else if (type == OT_NOISE) {
					const uint32_t nz = (f.type >> 8) & 0xff;
					const uint32_t n0 = f.phase0, nprev = f.prev_phase;
					if (SCAN == 2 && nz == NZ_re) {
						/* noise.h:136-147: sum += (int32_t)hash(n++) >> 6, wrapping; the sample is the folded sum. A running
						 * sum like a phase's: per-row DPP scans, what the groups before have added by look-back */
						const FastAux fa = load_aux_uniform(faux + si);
						uint32_t S[T];
#pragma unroll
						for (int k = 0; k < T; ++k) {
							const int t = t0 + k * (int)RS;
							const uint32_t inc = in_seg(t) ? (uint32_t)((int32_t)ranfast32(n0 + (uint32_t)t) >> 6) : 0u;
							S[k] = wave_incl_scan_dpp(inc);
						}
						uint32_t acc;
						if (look_own) {
							acc = first_group ? nprev : (uint32_t)carry[si];
						} else {
							uint32_t tot = 0;
#pragma unroll
							for (int k = 0; k < T; ++k)
								tot += (uint32_t)__builtin_amdgcn_readlane((int)S[k], 63) - lead32(S[k], k);
							acc = nprev + (look_lds ? lookback32<true>(lk_base + (size_t)fa.pad[0] * 2 * 64, cg, tot, 0, lk_ring, cgm, l, zero_acc)
							                        : lookback32<false>(lookv + (size_t)fa.pad[0] * 2 * P.scan_groups, cg, tot, P.look_epoch, 0, 0, l, zero_acc, (P.look_wpv_flags & 2u) != 0));
						}
#pragma unroll
						for (int k = 0; k < T; ++k) {
							const int t = t0 + k * (int)RS;
							const uint32_t lead = lead32(S[k], k);
							const uint32_t last = (uint32_t)__builtin_amdgcn_readlane((int)S[k], 63);
							const uint32_t sum = acc + (S[k] - lead);
							acc += last - lead;
							s[k] = fscalei((uint32_t)foldhd32((int32_t)sum), 0x1p-31f);
							/* the sum after the segment's last frame: the operator's next `prev` (finalize_kernel) */
							if (is_last_group && t == (int)fast_total - 1 && own(k)) P.ops[f.gop].st_prev_phase = sum;
						}
						if (look_own && l == 0) carry[si] = (unsigned long long)acc;
					} else {
#pragma unroll
					for (int k = 0; k < T; ++k) {
						const int t = t0 + k * (int)RS;
						const uint32_t n = n0 + (uint32_t)t;
						if (nz == NZ_vi) {
							uint32_t s1 = ranfast32(n);
							uint32_t s0 = (EDGE && t == 0) ? nprev : ranfast32(n - 1);
							s[k] = fscalei((s1 / 2) - (s0 / 2), 0x1p-31f);
						} else if (nz == NZ_bv) {
							int32_t s1 = noise_bv_term(n);
							int32_t s0 = (EDGE && t == 0) ? (int32_t)nprev : noise_bv_term(n - 1);
							s[k] = (float)(s1 - s0);
						} else {
							s[k] = noise_stateless(nz, n);
						}
					}
					}
				} else { /* OT_AMP (generator.c:517-518: 1), or an oscillator whose output stands still */
#pragma unroll
					for (int k = 0; k < T; ++k) s[k] = f.fc;
				}
				/* amplitude and combine: generator.c:384-440 */
				float r[T];
				if (f.amp_off != ~0u) {
#pragma unroll
					for (int k = 0; k < T; ++k) r[k] = slots[f.amp_off + k * 64];
				} else if (f.ramp & 1) { /* amplitude ramp in progress, sau/line.c:65-281 */
					const FastLine fl = load_line_uniform(flines + si);
#pragma unroll
					for (int k = 0; k < T; ++k) r[k] = fast_line_value(fl, t0 + k * (int)RS);
				} else {
#pragma unroll
					for (int k = 0; k < T; ++k) r[k] = f.ac;
				}
				if (layer) {
#pragma unroll
					for (int k = 0; k < T; ++k)
						r[k] = mix_combine(slots[f.out_off + k * 64], s[k], r[k], wave_env, true);
				} else if (wave_env) {
#pragma unroll
					for (int k = 0; k < T; ++k) r[k] = mix_combine(0.f, s[k], r[k], true, false);
				} else {
#pragma unroll
					for (int k = 0; k < T; ++k) r[k] = s[k] * r[k];
				}
				if (to_voice) {
#pragma unroll
					for (int k = 0; k < T; ++k) {
						const int t = t0 + k * (int)RS;
						const bool mine = REPAIR ? repair_mine(k) : own(k);
						if (mine && in_end(t)) FK_VSTORE(&vrow[t], r[k]);
					}
					if constexpr (TAIL && SCAN == 2 && !REPAIR && !CUB) {
						if (tail_n) {
							/* the stream's last row: mixed here, in the mixer's order and with its operations (k_finish.h: mix_few_kernel;
							 * generator.c:749-825) -- the rows before from HBM, eight loads in flight per row, then this one's samples */
							const MixStream ms = P.inmix_stream[tail_s];
							const float *row0 = P.vout + (size_t)ms.first_row * P.row_stride;
							const float pan_own = P.vinfo[vd.out_row].pan_const;
							const bool swap = (P.tail_flags & 2u) != 0;
							constexpr int TH = T >= 4 ? 4 : T; /* (rows at a time: what the sums and the loads in flight take of the registers) */
#pragma unroll
							for (int h = 0; h < T; h += TH) {
								float mL[TH], mR[TH];
#pragma unroll
								for (int k = 0; k < TH; ++k) { mL[k] = 0.f; mR[k] = 0.f; }
								for (uint32_t j = 0; j + 1 < tail_n; ++j) {
									/* (a row's address: a buffer descriptor in scalar registers + the frame's 32-bit offset -- as plain pointers the
									 * loads took a 64-bit address each, hoisted out of the step loop and spilled. The whole offset in the lane's
									 * register: the range check looks at that register alone, and a group's first lanes, whose row 0 lies before the
									 * segment, own frames of the rows after it. A lane before the segment reads outside the descriptor's range: 0) */
									const auto rs = __builtin_amdgcn_make_buffer_rsrc((void *)(row0 + (size_t)j * P.row_stride), 0, P.row_stride * 4u, 0x00020000);
									const float pan = P.vinfo[ms.first_row + j].pan_const;
									float x[TH];
#pragma unroll
									for (int k = 0; k < TH; ++k)
										x[k] = (h + k < T) ? __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rs, (t0 + (h + k) * (int)RS) * 4, 0, 2 /* nt */)) : 0.f;
#pragma unroll
									for (int k = 0; k < TH; ++k) {
										const float v = x[k] * ms.amp_scale;
										const float s_r = v * pan;
										mL[k] = (mL[k] + v) - s_r;
										mR[k] = (mR[k] + v) + s_r;
									}
								}
#pragma unroll
								for (int k = 0; k < TH; ++k) {
									if (h + k >= T) continue;
									const int t = t0 + (h + k) * (int)RS;
									const float v = r[h + k] * ms.amp_scale;
									const float s_r = v * pan_own;
									const float fl = (mL[k] + v) - s_r, fr = (mR[k] + v) + s_r;
									if (own(h + k) && in_end(t)) {
										if (P.tail_flags & 1u) {
											const int16_t l16 = pcm16(fl), r16 = pcm16(fr);
											const uint32_t lo = (uint16_t)(swap ? pcm_swap(l16) : l16), hi = (uint16_t)(swap ? pcm_swap(r16) : r16);
											*(u32_alias *)(ms.pcm + 2 * (size_t)(P.tail_pcm_offset + (uint32_t)t)) = lo | (hi << 16);
										} else {
											const int16_t m16 = pcm16((fl + fr) * 0.5f);
											ms.pcm[P.tail_pcm_offset + (uint32_t)t] = swap ? pcm_swap(m16) : m16;
										}
									}
								}
							}
						}
					}
				} else {
#pragma unroll
					for (int k = 0; k < T; ++k) slots[f.out_off + k * 64] = r[k];
				}
			} else if (kind == ST_LINE) {
				/* held line: v0 (sau/line.c:435-442); ratio lines only exist for freq */
				if (f.ramp) {
					FastLine fl;
					fl.goal_len = 0; fl.hold = f.ac; fl.pad = 0;
					fl.sw = sweep_setup(LN_sah, 0.f, 0.f, 0, 1);
					if (f.ramp & 1) fl = load_line_uniform(flines + si);
					uint32_t mflags = 0, fmul_off = ~0u;
					float mulc = 1.f;
					if (SCAN && (f.ramp & 2)) { /* ratio line: x the parent's frequency (sau/line.c:72) */
						const FastAux fa = load_aux_uniform(faux + si);
						mflags = fa.flags; fmul_off = fa.fmul_off; mulc = fa.mulc;
					}
#pragma unroll
					for (int k = 0; k < T; ++k) {
						const int t = t0 + k * (int)RS;
						float v = fast_line_value(fl, t);
						const bool in_goal = (uint32_t)t < fl.goal_len;
						if (mflags & (in_goal ? FA_MUL_GOAL : FA_MUL_HOLD))
							v *= fmul_off != ~0u ? slots[fmul_off + k * 64] : mulc;
						slots[f.out_off + k * 64] = v;
					}
				} else {
#pragma unroll
					for (int k = 0; k < T; ++k) slots[f.out_off + k * 64] = f.ac;
				}
			} else if (kind == ST_LERP) { /* generator.c:466-467 */
				/* (round 5: a range end that is one value for the segment has no block -- decode_kernel hands it over as f.fc) */
				const bool end_const = f.aux_off == ~0u;
#pragma unroll
				for (int k = 0; k < T; ++k) {
					float pv = slots[f.out_off + k * 64];
					pv += ((end_const ? f.fc : slots[f.aux_off + k * 64]) - pv) * slots[f.pm_off + k * 64];
					slots[f.out_off + k * 64] = pv;
				}
			} else if (kind == ST_VOICE) { /* generator.c:749-788 with pan modulators */
#pragma unroll
				for (int k = 0; k < T; ++k) {
					const int t = t0 + k * (int)RS;
					const bool mine = REPAIR ? repair_mine(k) : own(k);
					if (mine && in_end(t)) {
						FK_VSTORE(&vrow[t], slots[f.out_off + k * 64]);
						if (prow) prow[t] = f.pm_off != ~0u ? slots[f.pm_off + k * 64] : f.pan;
					}
				}
			}
		}
		if (held_rows || held_far) {
			/* to the repair pass -- unless this is it, the group touches an end of the segment
			 * (carried state sits at fixed lanes there) or the voice has running sums */
			bool noted = false;
			/* (a contiguous group's later rows resolve their holds through the rows before: one left there means a run that
			 * reaches back to the group's lead-in -- not a case for the one-frame repair) */
			if (!REPAIR && !SCAN && !CUB && P.repair_on && !first_group && !is_last_group && !held_far &&
			    (int)(cg * GF) - (int)H >= (int)FAST_REPAIR_SHIFT) {
				uint32_t at = 0;
				if (l == 0) at = atomicAdd(&rep[0], 1u);
				at = uni(at);
				if (at < FAST_MAX_REPAIR) {
					if (l == 0) {
						rep[2 + 2 * at] = cg;
						rep[3 + 2 * at] = held_rows;
						atomicOr(&P.pass_flags[FAST_MAX_LEVELS], 1u);
						atomicOr(&P.work_count[1], 2u); /* (frames the mixer may have taken early -- k_finish.h: premix_kernel -- change in the repair pass) */
					}
					noted = true;
				}
			}
			if (!noted) zero_acc = 1;
		}
