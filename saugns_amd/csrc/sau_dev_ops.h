/* sau_dev_ops.h -- operator state transitions that are not per-sample work:
 * event application (generator.c:245-343 prepare_op/update_op with the
 * setters of wosc.h:55-91, rasg.h:44-119, noise.h:29-36).  Host+device.
 */
#ifndef SAU_DEV_OPS_H
#define SAU_DEV_OPS_H

#include "sau_dev_types.h"

namespace saudev {

/* ---- R oscillator counter packing, rasg.h:59-92 ---------------------------- */
SAU_HD uint32_t ras_get_cycle(const DevOp &n) {
	return (uint32_t)(n.cycle_phase >> 32) & ~1u;
}
SAU_HD uint32_t ras_get_phase(const DevOp &n) {
	return (n.flags & OPF_RATE2X) ? (uint32_t)(n.cycle_phase >> 1) : (uint32_t)n.cycle_phase;
}
SAU_HD void ras_set_cycle(DevOp &n, uint32_t cycle) {
	uint32_t phase = ras_get_phase(n);
	uint64_t p64 = (n.flags & OPF_RATE2X) ? ((uint64_t)phase) << 1 : (uint64_t)phase;
	n.cycle_phase = ((uint64_t)(cycle & ~1u)) << 32 | p64;
}
SAU_HD void ras_set_phase(DevOp &n, uint32_t phase) {
	uint32_t cycle = ras_get_cycle(n);
	uint64_t p64 = (n.flags & OPF_RATE2X) ? ((uint64_t)phase) << 1 : (uint64_t)phase;
	n.cycle_phase = ((uint64_t)cycle) << 32 | p64;
}
/* rasg.h:97-119 */
SAU_HD void ras_set_opt(DevOp &n, const OpUpdate &u) {
	uint32_t flags = u.ras_flags;
	if (u.ras_flags & RO_LINE_SET) n.wave = u.ras_line;
	if (u.ras_flags & RO_FUNC_SET) n.ras_func = u.ras_func;
	else flags |= n.ras_flags;
	if (u.ras_flags & RO_LEVEL_SET) n.ras_level = u.ras_level;
	if (u.ras_flags & RO_ASUBVAL_SET) n.ras_alpha = u.ras_alpha;
	n.ras_flags = flags & 0x3ffu; /* 10-bit field in sauRasOpt */
	bool rate2x = !(flags & RO_HALFSHAPE);
	bool cur = (n.flags & OPF_RATE2X) != 0;
	if (rate2x != cur) {
		uint32_t cycle = ras_get_cycle(n);
		uint32_t phase = ras_get_phase(n);
		if (rate2x) n.flags |= OPF_RATE2X; else n.flags &= ~OPF_RATE2X;
		ras_set_cycle(n, cycle);
		ras_set_phase(n, phase);
	}
}

/* generator.c:245-343 */
SAU_HD void apply_update(DevOp &n, const OpUpdate &u, const WaveConst *wc) {
	if (u.first) {
		/* prepare_op: zeroed node, type-specific initial state */
		n = DevOp();
		n.type = u.type;
		n.coeff = u.coeff;
		if (u.type == OT_WAVE) {
			n.wave = 0; /* sin */
			n.phase = (uint32_t)wc[0].phase_adj;
			n.flags |= OPF_OSC_RESET;
		} else if (u.type == OT_RASEG) {
			n.flags |= OPF_RATE2X;
			n.wave = LN_lin;
			n.ras_func = RF_URAND;
			n.ras_level = ras_level9();
			n.ras_alpha = 0x9e3779b9u;
			n.ras_flags = 0;
		}
	}
	bool osc = false;
	switch (u.type) {
	case OT_NOISE:
		if (u.params & POPP_MODE) { n.wave = u.mode_main; n.noise_prev = 0; }
		if (u.params & POPP_SEED) n.noise_n = u.seed;
		break;
	case OT_WAVE:
		if (u.params & POPP_MODE) {
			uint32_t wave = u.mode_main < 12 ? u.mode_main : 0;
			n.phase += (uint32_t)wc[wave].phase_adj - (uint32_t)wc[n.wave].phase_adj;
			n.wave = wave;
			n.flags |= OPF_OSC_RESET;
		}
		if (u.params & POPP_PHASE)
			n.phase = u.phase + (uint32_t)wc[n.wave].phase_adj;
		osc = true;
		break;
	case OT_RASEG:
		if (u.params & POPP_MODE) ras_set_opt(n, u);
		if (u.params & POPP_PHASE) ras_set_phase(n, u.phase);
		if (u.params & POPP_SEED) ras_set_cycle(n, u.seed);
		osc = true;
		break;
	default: break;
	}
	if (osc) {
		line_copy(n.line[L_FREQ], u.line[L_FREQ], u.loop_tails != 0);
		line_copy(n.line[L_FREQ2], u.line[L_FREQ2], u.loop_tails != 0);
		line_copy(n.line[L_PMA], u.line[L_PMA], u.loop_tails != 0);
	}
	if (u.params & POPP_TIME) {
		n.time = u.time;
		if (u.time_inf) n.flags |= OPF_TIME_INF; else n.flags &= ~OPF_TIME_INF;
	}
	line_copy(n.line[L_AMP], u.line[L_AMP], u.loop_tails != 0);
	line_copy(n.line[L_AMP2], u.line[L_AMP2], u.loop_tails != 0);
	line_copy(n.line[L_PAN], u.line[L_PAN], u.loop_tails != 0);
}

} /* namespace saudev */
#endif
