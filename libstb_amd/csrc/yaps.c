/*
 * yaps.c -- message / fatal-error sink behind include/yaps.h.
 *
 * Behaviour follows the reference's lib/yaps.c:24-81: text goes to stderr unless the application
 * installed its own va_list sink with yaps_yapper(); yaps_quit and yaps_sysquit terminate the
 * process with exit(1); yaps_sysquit first reports strerror(errno) followed by ": ".
 */
#include "../../include/yaps.h"

#include <errno.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

static void (*sink)(const char *format, va_list ap) = NULL;

void yaps_yapper(void (*yapper)(const char *format, va_list ap)) { sink = yapper; }

static void emit(const char *fmt, va_list ap) {
  if (sink)
    sink(fmt, ap);
  else
    vfprintf(stderr, fmt, ap);
}

/* a sink only takes va_lists, so fixed text needs a variadic trampoline */
static void emitf(const char *fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  emit(fmt, ap);
  va_end(ap);
}

void yaps_message(const char *fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  emit(fmt, ap);
  va_end(ap);
}

void yaps_quit(const char *fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  emit(fmt, ap);
  va_end(ap);
  exit(1);
}

void yaps_sysquit(const char *fmt, ...) {
  va_list ap;
  emitf("%s: ", strerror(errno));
  va_start(ap, fmt);
  emit(fmt, ap);
  va_end(ap);
  exit(1);
}
