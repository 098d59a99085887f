/* synthhost.h -- host-side helpers of the synthplayer hot path in native code (no device code, no HIP): libsynthhost.so, built
 * by g++ from synthesizer_amd/csrc/host_tables.cpp (synthesizer_amd/build.py build_host()).
 *
 * The reference has no counterpart: its oscillators carry a float64 running sum `t += increment` through their blocks() loops
 * (synthplayer/oscillators.py, every blocks() preamble and loop; line numbers uncitable, the source is not mounted).  The device
 * evaluates that sum in closed form from a table of exactly linear pieces; building the table -- ~150 pieces
 * per distinct (phase, frequency) -- was most of what creating a bank cost on the host while it was a Python loop
 * (synthesizer_amd/phasetable.py build_phase_table, which stays as the tested statement of the algorithm and the fallback).
 */
#ifndef SYNTHHOST_H
#define SYNTHHOST_H
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

/* one piece: t_n = t0 + (n - n0) * dt exactly (in float64) for n0 <= n < the next piece's n0; the layout of sh_segment
 * (include/synthhip.h) and of _native.SEGMENT_DTYPE */
typedef struct shh_segment { uint64_t n0; double t0; double dt; } shh_segment;

/* Pieces of the sequence t_0 = t0, t_{n+1} = fl(t_n + inc) covering [0, n_limit), written to out[0 .. cap).
 * Returns the number of pieces; -1 when more than cap are needed (nothing useful in out); -2 for a sequence that does not close
 * (a denormal increment from a denormal start: 2^70 single steps). */
int shh_phase_table(double t0, double inc, uint64_t n_limit, shh_segment* out, int cap);

/* "synthhost <version> src:<16 hex digits of the source hash>" */
const char* shh_version(void);

#ifdef __cplusplus
}
#endif
#endif
