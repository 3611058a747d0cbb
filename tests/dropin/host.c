/* host.c -- a minimal C host in the shape of the reference's Player_run
 * (saugns.c:583-621): parse a script with the reference's own front end,
 * create a generator, call sauGenerator_run in 11289-frame chunks, write the
 * int16 PCM to stdout.  It is compiled twice by tests/test_gpu_dropin.py:
 *
 *   cc host.c -lsaugns_amd -lsau_ref   -> generator symbols resolve to this repo's GPU backend
 *   cc host.c -lsau_ref                -> the reference's CPU generator
 *
 * i.e. the same host source, only the link line differs.
 * The declarations below restate the interfaces the host uses:
 * sau/script.h:128-141, sau/program.h:269-270, sau/generator.h:17-26.
 * Test infrastructure; not part of the product. */
#include <stdbool.h>
#include <stddef.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>

typedef struct sauProgram sauProgram;
typedef struct sauGenerator sauGenerator;
typedef struct sauScriptPredef { const char *key; uint32_t len; double val; } sauScriptPredef;
typedef struct sauScriptArg {
	const char *str;
	bool is_path : 1;
	bool no_time : 1;
	sauScriptPredef *predef;
	size_t predef_count;
} sauScriptArg;

sauProgram *sau_build_Program(const sauScriptArg *arg);
void sau_discard_Program(sauProgram *o);
sauGenerator *sau_create_Generator(const sauProgram *prg, uint32_t srate);
void sau_destroy_Generator(sauGenerator *o);
bool sauGenerator_run(sauGenerator *o, int16_t *buf, size_t buf_len, bool stereo, size_t *out_len);

int main(int argc, char **argv) {
	if (argc < 3) { fprintf(stderr, "usage: host <script text> <mono|stereo> [srate]\n"); return 2; }
	const bool stereo = argv[2][0] == 's';
	const uint32_t srate = argc > 3 ? (uint32_t)atol(argv[3]) : 44100;
	sauScriptArg arg = {0};
	arg.str = argv[1];
	arg.no_time = true; /* deterministic seeds, as the reference's -d */
	sauProgram *prg = sau_build_Program(&arg);
	if (!prg) return 1;
	sauGenerator *gen = sau_create_Generator(prg, srate);
	if (!gen) { sau_discard_Program(prg); return 1; } /* saugns.c:583-588 */
	enum { BUF_LEN = 11289 };
	static int16_t buf[BUF_LEN * 2];
	bool run;
	do {
		size_t len = 0;
		run = sauGenerator_run(gen, buf, BUF_LEN, stereo, &len);
		fwrite(buf, sizeof(int16_t) * (stereo ? 2 : 1), len, stdout);
	} while (run);
	sau_destroy_Generator(gen);
	sau_discard_Program(prg);
	return 0;
}
