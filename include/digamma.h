/*
 * digamma.h -- compatibility header for callers that include the reference's lib/digamma.h
 * (e.g. test/demo.c:27).  As in the reference's shipped configuration (lib/digamma.h:25,
 * LS_NOPOLYGAMMA defined) only Radford Neal's digamma is available and `digamma(x)` maps to it
 * (lib/digamma.h:38-41); the polygamma family and digammaInv are not part of this library.
 */
#ifndef STB_AMD_DIGAMMA_H
#define STB_AMD_DIGAMMA_H
#ifndef __DIGAMMA_H
#define __DIGAMMA_H
#endif
#define LS_NOPOLYGAMMA
#include "sapprox.h" /* declares digammaRN */
#define digamma(x) digammaRN(x)
#endif
