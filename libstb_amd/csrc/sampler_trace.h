/* sampler_trace.h -- private: the last sampler call's log-posterior evaluations, for tests and
 * diagnostics (exported as stb_sampler_trace_* in include/stb_hip.h). */
#ifndef STB_SAMPLER_TRACE_H
#define STB_SAMPLER_TRACE_H
void stb_trace_reset(void);
void stb_trace_add(double x, double y);
void stb_trace_code(int code);
#endif
