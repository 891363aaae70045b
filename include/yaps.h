/*
 * yaps.h -- message / fatal-error sink; drop-in for the reference's lib/yaps.h:16-19.
 * Behaviour (lib/yaps.c:24-81): messages go to stderr unless a sink was installed with
 * yaps_yapper(); the *quit variants then exit(1); yaps_sysquit prefixes strerror(errno).
 */
#ifndef STB_AMD_YAPS_H
#define STB_AMD_YAPS_H
#include <stdarg.h>
#ifdef __cplusplus
extern "C" {
#endif
void yaps_message(const char *fmt, ...);                          /* lib/yaps.h:16 */
void yaps_quit(const char *fmt, ...);                             /* lib/yaps.h:17 */
void yaps_sysquit(const char *fmt, ...);                          /* lib/yaps.h:18 */
void yaps_yapper(void (*yapper)(const char *format, va_list ap)); /* lib/yaps.h:19 */
#ifdef __cplusplus
}
#endif
#endif
