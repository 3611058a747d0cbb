/* sau_abi.h -- binary contract between a SAU host (parser/player) and a
 * generator backend, as laid out on x86-64 SysV.
 *
 * This header is written for this repository; it re-declares, field for
 * field, the in-memory layout that the reference host hands to its
 * generator so that the MI355X backend can be linked in place of
 * sau/generator.o.  Every struct carries compile-time offset checks
 * (values measured against the reference headers).
 *
 * Reference interfaces mirrored (file:line in saugns v0.4.7):
 *   sauLine .................. sau/line.h:99-121
 *   sauTime .................. sau/program.h:25-39
 *   sauRasOpt / mode union ... sau/program.h:126-163,224-227
 *   sauProgramIDArr .......... sau/program.h:177-180
 *   sauProgramOpRef .......... sau/program.h:206-210
 *   sauProgramOpData ......... sau/program.h:212-231
 *   sauProgramEvent .......... sau/program.h:233-241
 *   sauProgram ............... sau/program.h:253-265
 */
#ifndef SAU_ABI_H
#define SAU_ABI_H

#include <stddef.h>
#include <stdint.h>
#include <stdbool.h>

#ifdef __cplusplus
extern "C" {
#define SAU_ABI_ASSERT(c, m) static_assert(c, m)
#else
#define SAU_ABI_ASSERT(c, m) _Static_assert(c, m)
#endif

/* ---- value lines (ramps / envelopes) ---------------------------------- */

/* shape ids, in table order (sau/line.h:18-32) */
enum {
	SAU_LINE_N_cos = 0, SAU_LINE_N_lin, SAU_LINE_N_sah, SAU_LINE_N_exp,
	SAU_LINE_N_log, SAU_LINE_N_xpe, SAU_LINE_N_lge, SAU_LINE_N_sqe,
	SAU_LINE_N_cub, SAU_LINE_N_smo, SAU_LINE_N_ncl, SAU_LINE_N_nhl,
	SAU_LINE_N_uwh,
	SAU_LINE_NAMED
};

/* sauLine.flags bits (sau/line.h:99-107) */
enum {
	SAU_LINEP_STATE       = 1 << 0, /* v0 given */
	SAU_LINEP_STATE_RATIO = 1 << 1, /* v0 is a ratio of the parent frequency */
	SAU_LINEP_GOAL        = 1 << 2, /* vt given; a timed sweep is pending */
	SAU_LINEP_GOAL_RATIO  = 1 << 3,
	SAU_LINEP_TYPE        = 1 << 4, /* shape given */
	SAU_LINEP_TIME        = 1 << 5, /* time given; cleared once it has run out */
	SAU_LINEP_TIME_IF_NEW = 1 << 6  /* keep a still-running time */
};

typedef struct sauLine {
	float v0, vt;        /* state value, goal value */
	uint32_t pos, end;   /* sweep position and length, in samples */
	uint32_t time_ms;
	uint8_t type;        /* SAU_LINE_N_* */
	uint8_t flags;       /* SAU_LINEP_* */
} sauLine;
SAU_ABI_ASSERT(sizeof(sauLine) == 24, "sauLine size");
SAU_ABI_ASSERT(offsetof(sauLine, type) == 20 && offsetof(sauLine, flags) == 21,
		"sauLine tail");

/* ---- time ------------------------------------------------------------- */

enum {
	SAU_TIMEP_SET      = 1 << 0,
	SAU_TIMEP_DEFAULT  = 1 << 1,
	SAU_TIMEP_IMPLICIT = 1 << 2  /* lasts as long as whatever uses it */
};

typedef struct sauTime {
	uint32_t v_ms;
	uint8_t flags;
} sauTime;
SAU_ABI_ASSERT(sizeof(sauTime) == 8, "sauTime size");

/* ---- operator kinds, wave / noise ids --------------------------------- */

enum { /* sauProgramOpData.type (sau/program.h:69-80) */
	SAU_POPT_N_amp = 0, /* 'A' */
	SAU_POPT_N_noise,   /* 'N' */
	SAU_POPT_N_wave,    /* 'W' */
	SAU_POPT_N_raseg,   /* 'R' */
	SAU_POPT_TYPES
};

enum { /* sauProgramOpData.params bits (sau/program.h:93-99) */
	SAU_POPP_TIME  = 1 << 0,
	SAU_POPP_MODE  = 1 << 1,
	SAU_POPP_PHASE = 1 << 2,
	SAU_POPP_SEED  = 1 << 3,
	SAU_POP_PARAMS = (1 << 4) - 1
};

enum { /* wave ids, table order (sau/wave.h:33-81) */
	SAU_WAVE_N_sin = 0, SAU_WAVE_N_tri, SAU_WAVE_N_srs, SAU_WAVE_N_sqr,
	SAU_WAVE_N_ean, SAU_WAVE_N_cat, SAU_WAVE_N_eto, SAU_WAVE_N_par,
	SAU_WAVE_N_mto, SAU_WAVE_N_saw, SAU_WAVE_N_hsi, SAU_WAVE_N_spa,
	SAU_WAVE_NAMED
};

enum { /* noise ids (sau/program.h:102-120) */
	SAU_NOISE_N_wh = 0, SAU_NOISE_N_gw, SAU_NOISE_N_bw, SAU_NOISE_N_tw,
	SAU_NOISE_N_re, SAU_NOISE_N_vi, SAU_NOISE_N_bv,
	SAU_NOISE_NAMED
};

/* ---- random-segments ('R') options ------------------------------------ */

enum { /* sauRasOpt.func (sau/program.h:135-143) */
	SAU_RAS_F_URAND = 0, SAU_RAS_F_GAUSS, SAU_RAS_F_BIN, SAU_RAS_F_TERN,
	SAU_RAS_F_FIXED, SAU_RAS_F_ADDREC,
	SAU_RAS_FUNCTIONS
};

enum { /* sauRasOpt.flags (sau/program.h:151-163) */
	SAU_RAS_O_PERLIN      = 1u << 0,
	SAU_RAS_O_HALFSHAPE   = 1u << 1,
	SAU_RAS_O_ZIGZAG      = 1u << 2,
	SAU_RAS_O_SQUARE      = 1u << 3,
	SAU_RAS_O_VIOLET      = 1u << 4,
	SAU_RAS_O_UNUSED      = 1u << 5,
	SAU_RAS_O_FUNC_FLAGS  = (1u << 6) - 1,
	SAU_RAS_O_LINE_SET    = 1u << 6,
	SAU_RAS_O_FUNC_SET    = 1u << 7,
	SAU_RAS_O_LEVEL_SET   = 1u << 8,
	SAU_RAS_O_ASUBVAL_SET = 1u << 9
};

typedef struct sauRasOpt {
	uint8_t line;        /* byte 0: line shape */
	unsigned flags : 10; /* bits 8..17 of the first word */
	unsigned func  : 6;  /* bits 18..23 */
	unsigned level : 8;  /* byte 3 */
	uint32_t alpha;
} sauRasOpt;
SAU_ABI_ASSERT(sizeof(sauRasOpt) == 8, "sauRasOpt size");

typedef union sauPOPMode {
	uint8_t main;  /* wave id or noise id */
	sauRasOpt ras;
} sauPOPMode;

/* ---- id lists, graph refs --------------------------------------------- */

#define SAU_PVO_NO_ID UINT16_MAX
#define SAU_POP_NO_ID UINT32_MAX

typedef struct sauProgramIDArr {
	uint32_t count;
	uint32_t ids[];
} sauProgramIDArr;

enum { /* how an operator is used (sau/program.h:183-204) */
	SAU_POP_N_carr = 0, SAU_POP_N_camod, SAU_POP_N_amod, SAU_POP_N_ramod,
	SAU_POP_N_fmod, SAU_POP_N_rfmod, SAU_POP_N_pmod, SAU_POP_N_apmod,
	SAU_POP_N_fpmod,
	SAU_POP_NAMED
};

typedef struct sauProgramOpRef {
	uint32_t id;
	uint8_t use;
	uint8_t level;
} sauProgramOpRef;
SAU_ABI_ASSERT(sizeof(sauProgramOpRef) == 8, "sauProgramOpRef size");

/* ---- per-event operator update ---------------------------------------- */

typedef struct sauProgramOpData {
	uint32_t id;
	uint32_t params;               /* SAU_POPP_* */
	sauTime time;
	sauLine *pan;                  /* NULL pointers mean "unchanged" */
	sauLine *amp, *amp2;
	sauLine *freq, *freq2;
	sauLine *pm_a;
	uint32_t phase;                /* cycle fraction, 2^32 = one turn */
	uint32_t seed;
	uint8_t use_type;              /* SAU_POP_N_* */
	uint8_t type;                  /* SAU_POPT_N_* */
	sauPOPMode mode;
	const sauProgramIDArr *camods; /* pan modulators */
	const sauProgramIDArr *amods, *ramods;
	const sauProgramIDArr *fmods, *rfmods;
	const sauProgramIDArr *pmods, *apmods, *fpmods;
} sauProgramOpData;
SAU_ABI_ASSERT(sizeof(sauProgramOpData) == 152, "sauProgramOpData size");
SAU_ABI_ASSERT(offsetof(sauProgramOpData, pan) == 16, "opdata.pan");
SAU_ABI_ASSERT(offsetof(sauProgramOpData, pm_a) == 56, "opdata.pm_a");
SAU_ABI_ASSERT(offsetof(sauProgramOpData, phase) == 64, "opdata.phase");
SAU_ABI_ASSERT(offsetof(sauProgramOpData, use_type) == 72, "opdata.use_type");
SAU_ABI_ASSERT(offsetof(sauProgramOpData, mode) == 76, "opdata.mode");
SAU_ABI_ASSERT(offsetof(sauProgramOpData, camods) == 88, "opdata.camods");
SAU_ABI_ASSERT(offsetof(sauProgramOpData, fpmods) == 144, "opdata.fpmods");

typedef struct sauProgramEvent {
	uint32_t wait_ms;
	uint16_t vo_id;
	uint32_t carr_op_id;
	uint32_t op_count;
	uint32_t op_data_count;
	const sauProgramOpRef *op_list;
	const sauProgramOpData *op_data;
} sauProgramEvent;
SAU_ABI_ASSERT(sizeof(sauProgramEvent) == 40, "sauProgramEvent size");
SAU_ABI_ASSERT(offsetof(sauProgramEvent, op_list) == 24, "event.op_list");

enum { SAU_PMODE_AMP_DIV_VOICES = 1 << 0 };

typedef struct sauProgram {
	const sauProgramEvent *events;
	size_t ev_count;
	uint16_t mode;
	uint16_t vo_count;
	uint32_t op_count;
	uint8_t op_nest_depth;
	uint32_t duration_ms;
	float ampmult;
	const char *name;
	void *mp;     /* host's memory pool; opaque here */
	void *parse;  /* host's parse result; opaque here */
} sauProgram;
SAU_ABI_ASSERT(sizeof(sauProgram) == 64, "sauProgram size");
SAU_ABI_ASSERT(offsetof(sauProgram, op_count) == 20, "program.op_count");
SAU_ABI_ASSERT(offsetof(sauProgram, ampmult) == 32, "program.ampmult");

#ifdef __cplusplus
}
#endif
#endif /* SAU_ABI_H */
