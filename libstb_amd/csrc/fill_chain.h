// fill_chain.h -- what the chain-form kernels share: trip and ring sizes, the kernel arguments, the
// LDS counter idiom and the bounded wait.
#ifndef STB_FILL_CHAIN_H
#define STB_FILL_CHAIN_H

#include "stb_common.h"

#define CH_U 8    // rows per trip
#define CH_RD 8   // trips in the significand ring (power of two)
#define CH_RE 32  // trips in the edge ring (power of two, >= 2 * 16)
#define CH_EOFF (1ull << 40)
#define CH_NEGZERO 0x8000000000000000ull

struct chain_args {
  unsigned *hdr;               // [0] ticket, [1] error code, [2] error detail; zeroed per fill
  unsigned long long *edge_v;  // [D][B][EV]  last column of a block, indexed by row
  unsigned long long *edge_e;  // [D][B][NP]  its lane exponent, indexed by trip, + CH_EOFF
  uint64_t EV, NP;
  int D, B;                    // tables, column blocks per table
  int TP, G;                   // trips per period, trips in all (rows 3 .. 2 + G*CH_U)
  unsigned long long timeout;  // wall_clock64 ticks a wait may last
  unsigned tp_magic;           // ceil(2^32 / TP): trip / TP = umulhi(trip, tp_magic) for trip < 2^26
  const unsigned *cnt;         // DOT = 1: occurrence count per cell, in the table's own layout
  const unsigned *item_ptr;    // DOT = 2: [trips * nsg + 1] first entry of every (trip, slice) item
  const unsigned short *ent_pos;  // DOT = 2: row-in-trip << 6 | column-in-slice of each occurring cell
  const unsigned *ent_cnt;     // DOT = 2: its occurrence count
  unsigned nsg;                // DOT = 2: slices per trip in item_ptr
  double *dotp;                // DOT kernels: [D][B][NC] partial sums of count * log S
  int poll_nap;                // s_sleep argument (64 cycles each) between two polls of a fetcher
  unsigned long long *dbg;     // diagnostic builds (make stamp): hand-off timeline [block<160][trip<1280][4]
  int mode;                    // diagnostic builds: 1 consumers only publish and release, 2 no table look-up,
                               // 3 the producer does not write the ring (results are wrong in all three)
};

__device__ __forceinline__ int lds_peek(const int *p) {
  return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
}
// LDS executes one wave's instructions in order: data written before the counter is visible to
// whoever sees the counter.  The empty asm keeps the compiler from reordering around it.
__device__ __forceinline__ void lds_post(int *p, int v) {
  asm volatile("" ::: "memory");
  if ((threadIdx.x & 63) == 0) __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
  asm volatile("" ::: "memory");
}

// the cold part of a bounded wait on an LDS counter, out of line so that the callers' row loops stay
// straight-line code: returns false when the wait was given up (timeout, or another wave gave up).
// The counter is polled every ~64 * nap cycles; the wall clock (an s_memrealtime round trip through
// the scalar cache, hundreds of cycles) and the abort flag are looked at once per 64 polls only, so
// that a waiter reacts within a fraction of a microsecond.
__device__ __attribute__((noinline)) bool chain_wait_slow(const int *cnt, int need, int *abort_flag, unsigned *hdr,
                                                          unsigned long long timeout, unsigned code, unsigned who, int nap) {
  unsigned long long t_begin = 0;
  bool timing = false;
  for (;;) {
    for (int k = 0; k < 64; k++) {
      // (a spinning wave takes issue slots from the producer it shares a SIMD with)
#ifdef STB_WAIT_NAP
      for (int i = 0; i < nap * STB_WAIT_NAP; i++) __builtin_amdgcn_s_sleep(1);
#else
      for (int i = 0; i < nap; i++) __builtin_amdgcn_s_sleep(1);
#endif
      if (__hip_atomic_load(cnt, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) >= need) return true;
    }
    if (__hip_atomic_load(abort_flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP)) return false;
    const unsigned long long now = wall_clock64();
    if (!timing) {
      timing = true;
      t_begin = now;
      if (timeout != 0) continue;
    }
    if (now - t_begin >= timeout) {
      if ((threadIdx.x & 63) == 0) {
        __hip_atomic_store(abort_flag, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        if (__hip_atomic_load(hdr + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == 0) {
          __hip_atomic_store(hdr + 2, who, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          __hip_atomic_store(hdr + 1, code, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
      }
      return false;
    }
  }
}

#endif
