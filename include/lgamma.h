/*
 * lgamma.h -- placeholder so that sources which include the reference's lib/lgamma.h still
 * compile (e.g. test/demo.c:26, lib/psample.h:20).  The gcache_* / pcache_* / qcache_* helpers
 * and gammadiff / psidiff declared by the reference's header (lib/lgamma.h:29-37) serve only the
 * compiled-out SAMPLEA_M variant (lib/samplea.c:92-139) and are NOT provided: SURVEY section 2,
 * component 8, out of scope.  Nothing is declared here.
 */
#ifndef STB_AMD_LGAMMA_H
#define STB_AMD_LGAMMA_H
#endif
