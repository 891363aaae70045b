// abi.hip -- the plain-C plumbing of include/stb_hip.h: error text, device selection, the buffer
// cache, memory and stream helpers, table layout queries.  No kernels here.
//
// No CPU fallback exists in this library: without a device every entry point fails with a message.

#include <mutex>
#include <vector>

#include "stb_common.h"

// ------------------------------------------------------------------------------------------------
// error plumbing

static thread_local char g_err[512] = "";

int stb_fail(const char *fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
  return 1;
}

extern "C" const char *stb_last_error(void) { return g_err; }

// ------------------------------------------------------------------------------------------------
// A GPU that is not ours alone.  The one-launch forms (halo blocks, grid, chain) have workgroups that wait for other
// workgroups of the same launch; that is safe on any device (a waiter only ever waits for a workgroup with a smaller
// ticket, which is resident or done) but not cheap on a shared one: when other processes' kernels hold the compute units
// the waiters cannot all be resident at once, each spins through whole scheduler quanta, and a 1.4 ms evaluation was seen
// to take 725 ms (four processes x 16 discounts on one MI355X, MEASUREMENTS section R6.3) -- with no time-out firing,
// because every single wait ends.  So: every such launch stamps its start and (its spine's) end, the host compares the
// span with what the geometry should take, counts the launches that took more than 20 x that (stb_slow_launches) and
// says so once; after two of them -- or at once with STB_SHARED_GPU=1 -- fills go to the producer/consumer form and
// evaluations through stored tables and the gather, which have no waits between workgroups and degrade in proportion.
// STB_SHARED_GPU=0 never switches.  The results of either route are within 1e-10 of the reference; they are not the
// same bits (a fused evaluation sums in the order of its tiles, the gather in the order of the sorted pairs).
#include <atomic>
static std::atomic<unsigned> g_slow_launches{0};
static std::atomic<int> g_shared_mode{-2};  // -2: not read from the environment yet; -1: automatic; 0 / 1: fixed
extern "C" void yaps_message(const char *fmt, ...);

static int shared_mode_now() {
  int m = g_shared_mode.load(std::memory_order_relaxed);
  if (m == -2) {
    m = stb_env_int("STB_SHARED_GPU", -1);
    if (m < -1 || m > 1) m = -1;
    g_shared_mode.store(m, std::memory_order_relaxed);
  }
  return m;
}

bool stb_shared_gpu() {
  const int m = shared_mode_now();
  if (m >= 0) return m == 1;
  return g_slow_launches.load(std::memory_order_relaxed) >= (unsigned)stb_env_int("STB_SHARED_AFTER", 2);
}

void stb_note_span(double span_ms, double expect_ms, const char *what) {
  if (!(span_ms > 0.0) || !(expect_ms > 0.0)) return;
  if (span_ms <= 20.0 * expect_ms || span_ms < 2.0) return;
  const unsigned n = g_slow_launches.fetch_add(1u) + 1u;
  if (n == 1u)
    yaps_message("libstb_amd: %s took %.1f ms on the device where about %.2f ms are expected -- the GPU is shared or throttled; "
                 "after %d such launches the library takes the forms without waits between workgroups (STB_SHARED_GPU=1 does so "
                 "from the start, =0 never)\n", what, span_ms, expect_ms, stb_env_int("STB_SHARED_AFTER", 2));
}

extern "C" unsigned stb_slow_launches(void) { return g_slow_launches.load(); }
extern "C" int stb_shared_gpu_mode(void) { return stb_shared_gpu() ? 1 : 0; }
// mode: 1 on, 0 off, -1 automatic (and the count of slow launches starts again)
extern "C" void stb_set_shared_gpu(int mode) {
  g_shared_mode.store(mode < -1 || mode > 1 ? -1 : mode);
  if (mode < 0) g_slow_launches.store(0u);
}
// (what the launches report; exported so that the policy can be exercised without a crowded GPU)
extern "C" void stb_note_launch_span(double span_ms, double expect_ms) { stb_note_span(span_ms, expect_ms, "a launch"); }

// ------------------------------------------------------------------------------------------------
// rand() guard (see stb_common.h)

// The lock covers only the count and the hand-over of the state: entry points of different threads run
// side by side (one host thread per GPU, or two threads on one GPU, overlap their device work), the
// caller's state goes away with the first one in and comes back with the last one out.
static std::mutex g_rand_mu;
static int g_rand_depth = 0;
static char g_rand_buf[128];
static char *g_rand_old = nullptr;

stb_rand_guard::stb_rand_guard() {
  std::lock_guard<std::mutex> lk(g_rand_mu);
  if (g_rand_depth++ == 0) g_rand_old = initstate(0x5eedu, g_rand_buf, sizeof g_rand_buf);
}
stb_rand_guard::~stb_rand_guard() {
  std::lock_guard<std::mutex> lk(g_rand_mu);
  if (--g_rand_depth == 0 && g_rand_old) setstate(g_rand_old);
}

// ------------------------------------------------------------------------------------------------
// devices.  A thread picks the GPU for the objects it creates with stb_set_device() (or the process
// does with STB_DEVICE=k); otherwise the HIP runtime's current device is used, which is what a
// torch caller has already set.  Tables and group sets remember their device and switch to it for
// the duration of every later call (stb_device_enter / stb_device_leave).

static thread_local int g_dev_choice = -1;

extern "C" int stb_device_count(void) {
  STB_ENTRY;
  int n = 0;
  if (hipGetDeviceCount(&n) != hipSuccess) {
    (void)hipGetLastError();
    return 0;
  }
  return n;
}

extern "C" int stb_set_device(int dev) {
  STB_ENTRY;
  int n = 0;
  if (hipGetDeviceCount(&n) != hipSuccess) {
    (void)hipGetLastError();
    n = 0;
  }
  if (dev < 0 || dev >= n) return stb_fail("stb_set_device: device %d of %d", dev, n);
  HIPCHK(hipSetDevice(dev));
  g_dev_choice = dev;
  return 0;
}

extern "C" int stb_get_device(void) {
  STB_ENTRY;
  if (g_dev_choice >= 0) return g_dev_choice;
  const char *s = getenv("STB_DEVICE");
  if (s && *s) return atoi(s);
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess) {
    (void)hipGetLastError();
    return -1;
  }
  return dev;
}

int stb_use_device(void) {
  const int want = stb_get_device();
  int cur = -1;
  if (want < 0 || hipGetDevice(&cur) != hipSuccess) {
    (void)hipGetLastError();
    return -1;
  }
  if (cur != want && hipSetDevice(want) != hipSuccess) {
    stb_fail("device %d cannot be selected: %s", want, hipGetErrorString(hipGetLastError()));
    return -1;
  }
  return want;
}

extern "C" int stb_device_enter(int dev) {
  STB_ENTRY;
  int cur = -1;
  if (hipGetDevice(&cur) != hipSuccess) {
    (void)hipGetLastError();
    return -1;
  }
  if (dev >= 0 && dev != cur) {
    if (hipSetDevice(dev) != hipSuccess) {
      (void)hipGetLastError();
      return -1;
    }
    return cur;
  }
  return -1;  // nothing to undo
}

extern "C" void stb_device_leave(int prev) {
  STB_ENTRY;
  if (prev >= 0) (void)hipSetDevice(prev);
}

extern "C" int stb_device_name(char *buf, int len) {
  STB_ENTRY;
  hipDeviceProp_t p;
  int dev = 0;
  HIPCHK(hipGetDevice(&dev));
  HIPCHK(hipGetDeviceProperties(&p, dev));
  snprintf(buf, len, "%s (%s, %d CUs)", p.name, p.gcnArchName, p.multiProcessorCount);
  return 0;
}

// ------------------------------------------------------------------------------------------------
// A small cache of device (and pinned host) buffers.  samplea builds and frees a group set (a dozen
// buffers, 70 MB at 10^6 pairs) on every call, as the reference builds and frees its table
// (lib/samplea.c:57-60,223); hipMalloc / hipFree cost ~1 ms each way there.  Freed buffers are kept
// (device memory up to STB_POOL_MB, default 4096; pinned host memory up to STB_POOL_HOST_MB,
// default 1024) and handed out again to requests of about the same size on the same device.
// Callers return a buffer only after the work that used it has completed.
struct pool_block {
  void *p;
  size_t bytes;
  int dev;
  int kind;  // 0: device memory, 1: pinned host memory
  bool used;
};
static std::mutex g_pool_mu;
static std::vector<pool_block> g_pool;
static size_t g_pool_idle[2] = {0, 0};

static hipError_t pool_raw_alloc(void **p, size_t bytes, int kind) {
  return kind ? hipHostMalloc(p, bytes, hipHostMallocDefault) : hipMalloc(p, bytes);
}
static void pool_raw_free(void *p, int kind) {
  if (kind) (void)hipHostFree(p);
  else (void)hipFree(p);
}

hipError_t stb_pool_malloc(void **out, size_t bytes, int kind) {
  if (bytes == 0) bytes = 1;
  int dev = 0;
  (void)hipGetDevice(&dev);
  {
    std::lock_guard<std::mutex> lk(g_pool_mu);
    int best = -1;
    for (size_t i = 0; i < g_pool.size(); i++) {
      const pool_block &b = g_pool[i];
      if (!b.used && b.dev == dev && b.kind == kind && b.bytes >= bytes && b.bytes <= bytes + bytes / 4 + 4096 &&
          (best < 0 || b.bytes < g_pool[best].bytes))
        best = (int)i;
    }
    if (best >= 0) {
      g_pool[best].used = true;
      g_pool_idle[kind] -= g_pool[best].bytes;
      *out = g_pool[best].p;
      return hipSuccess;
    }
  }
  void *p = nullptr;
  hipError_t e = pool_raw_alloc(&p, bytes, kind);
  if (e != hipSuccess) {
    // out of memory: give the idle buffers back and try once more
    std::vector<pool_block> drop;
    {
      std::lock_guard<std::mutex> lk(g_pool_mu);
      for (size_t i = 0; i < g_pool.size();) {
        if (!g_pool[i].used) {
          drop.push_back(g_pool[i]);
          g_pool_idle[g_pool[i].kind] -= g_pool[i].bytes;
          g_pool.erase(g_pool.begin() + i);
        } else {
          i++;
        }
      }
    }
    for (const pool_block &q : drop) pool_raw_free(q.p, q.kind);
    (void)hipGetLastError();
    e = pool_raw_alloc(&p, bytes, kind);
    if (e != hipSuccess) return e;
  }
  std::lock_guard<std::mutex> lk(g_pool_mu);
  g_pool.push_back(pool_block{p, bytes, dev, kind, true});
  *out = p;
  return hipSuccess;
}

bool stb_pool_free(void *p) {
  if (!p) return true;
  static const size_t cap[2] = {(size_t)stb_env_int("STB_POOL_MB", 4096) << 20,
                                (size_t)stb_env_int("STB_POOL_HOST_MB", 1024) << 20};
  int release = -1;  // kind to release with, -1: kept or unknown
  bool known = false;
  {
    std::lock_guard<std::mutex> lk(g_pool_mu);
    for (size_t i = 0; i < g_pool.size(); i++) {
      if (g_pool[i].p == p) {
        known = true;
        const int kind = g_pool[i].kind;
        if (g_pool_idle[kind] + g_pool[i].bytes <= cap[kind]) {
          g_pool[i].used = false;
          g_pool_idle[kind] += g_pool[i].bytes;
        } else {
          release = kind;
          g_pool.erase(g_pool.begin() + i);
        }
        break;
      }
    }
  }
  if (release >= 0) pool_raw_free(p, release);
  return known;
}

extern "C" void stb_pool_trim(void) {
  STB_ENTRY;
  std::vector<pool_block> drop;
  {
    std::lock_guard<std::mutex> lk(g_pool_mu);
    for (size_t i = 0; i < g_pool.size();) {
      if (!g_pool[i].used) {
        drop.push_back(g_pool[i]);
        g_pool.erase(g_pool.begin() + i);
      } else {
        i++;
      }
    }
    g_pool_idle[0] = g_pool_idle[1] = 0;
  }
  for (const pool_block &q : drop) pool_raw_free(q.p, q.kind);
}

extern "C" void *stb_device_malloc(size_t bytes) {
  STB_ENTRY;
  void *p = nullptr;
  hipError_t e = stb_pool_malloc(&p, bytes ? bytes : 1, 0);
  if (e != hipSuccess) {
    stb_fail("hipMalloc(%zu) failed: %s", bytes, hipGetErrorString(e));
    (void)hipGetLastError();
    return nullptr;
  }
  return p;
}
extern "C" void stb_device_free(void *p) {
  STB_ENTRY;
  if (p && !stb_pool_free(p)) (void)hipFree(p);
}
extern "C" void *stb_host_malloc(size_t bytes) {
  STB_ENTRY;
  void *p = nullptr;
  hipError_t e = stb_pool_malloc(&p, bytes ? bytes : 1, 1);
  if (e != hipSuccess) {
    stb_fail("hipHostMalloc(%zu) failed: %s", bytes, hipGetErrorString(e));
    (void)hipGetLastError();
    return nullptr;
  }
  return p;
}
extern "C" void stb_host_free(void *p) {
  STB_ENTRY;
  if (p && !stb_pool_free(p)) (void)hipHostFree(p);
}
// Pinned host memory on transparent huge pages: 2 MB-aligned, advised, then registered with the runtime.  What the host
// mirror of a table lives in: a caller's look-ups are random reads over hundreds of megabytes, and on the 4 KB pages of
// hipHostMalloc every one of them misses the TLB as well as the cache (10^6 random S_S calls on a 400 MB table: 27 ms
// against the reference's 15 on malloc'd memory, which the kernel backs with huge pages).  NULL when the runtime refuses.
#include <sys/mman.h>
extern "C" void *stb_host_malloc_huge(size_t bytes) {
  STB_ENTRY;
  const size_t A = (size_t)2 << 20, sz = ((bytes ? bytes : 1) + A - 1) & ~(A - 1);
  void *p = aligned_alloc(A, sz);
  if (!p) return nullptr;
  (void)madvise(p, sz, MADV_HUGEPAGE);
  if (hipHostRegister(p, sz, hipHostRegisterDefault) != hipSuccess) {
    (void)hipGetLastError();
    free(p);
    return nullptr;
  }
  return p;
}
extern "C" void stb_host_free_huge(void *p) {
  STB_ENTRY;
  if (!p) return;
  (void)hipHostUnregister(p);
  free(p);
}
extern "C" int stb_memcpy_h2d(void *d_dst, const void *h_src, size_t bytes, void *stream) {
  STB_ENTRY;
  HIPCHK(hipMemcpyAsync(d_dst, h_src, bytes, hipMemcpyHostToDevice, (hipStream_t)stream));
  return 0;
}
extern "C" int stb_memcpy_d2h(void *h_dst, const void *d_src, size_t bytes, void *stream) {
  STB_ENTRY;
  HIPCHK(hipMemcpyAsync(h_dst, d_src, bytes, hipMemcpyDeviceToHost, (hipStream_t)stream));
  return 0;
}
extern "C" int stb_stream_sync(void *stream) {
  STB_ENTRY;
  HIPCHK(hipStreamSynchronize((hipStream_t)stream));
  return 0;
}

// streams and events for the C host side (stable_host.c's look-ahead copies) and for FFI callers without HIP headers
extern "C" void *stb_stream_create(void) {
  STB_ENTRY;
  hipStream_t st = nullptr;
  if (hipStreamCreateWithFlags(&st, hipStreamNonBlocking) != hipSuccess) {
    stb_fail("hipStreamCreate: %s", hipGetErrorString(hipGetLastError()));
    return nullptr;
  }
  return (void *)st;
}
extern "C" void stb_stream_destroy(void *stream) {
  STB_ENTRY;
  if (stream) (void)hipStreamDestroy((hipStream_t)stream);
}
extern "C" void *stb_event_create(void) {
  STB_ENTRY;
  hipEvent_t ev = nullptr;
  if (hipEventCreateWithFlags(&ev, hipEventDisableTiming) != hipSuccess) {
    stb_fail("hipEventCreate: %s", hipGetErrorString(hipGetLastError()));
    return nullptr;
  }
  return (void *)ev;
}
extern "C" void stb_event_destroy(void *ev) {
  STB_ENTRY;
  if (ev) (void)hipEventDestroy((hipEvent_t)ev);
}
extern "C" int stb_event_record(void *ev, void *stream) {
  STB_ENTRY;
  HIPCHK(hipEventRecord((hipEvent_t)ev, (hipStream_t)stream));
  return 0;
}
extern "C" int stb_event_wait(void *ev) {
  STB_ENTRY;
  HIPCHK(hipEventSynchronize((hipEvent_t)ev));
  return 0;
}
/* 1 when everything recorded before the event is done, 0 when not yet, -1 on error */
extern "C" int stb_event_done(void *ev) {
  STB_ENTRY;
  const hipError_t e = hipEventQuery((hipEvent_t)ev);
  if (e == hipSuccess) return 1;
  (void)hipGetLastError();
  if (e == hipErrorNotReady) return 0;
  stb_fail("hipEventQuery: %s", hipGetErrorString(e));
  return -1;
}

extern "C" uint64_t stb_cells(unsigned N, unsigned M) { return stb_table_cells(N, M); }
extern "C" uint64_t stb_elems(unsigned N, unsigned M) { return stb_table_elems(N, M); }
extern "C" uint64_t stb_rowoff(unsigned n, unsigned M) { return stb_row_offset(n, M); }
extern "C" uint64_t stb_vcells(unsigned N, unsigned M) { return stb_vtable_cells(N, M); }
extern "C" uint64_t stb_velems(unsigned N, unsigned M) { return stb_vtable_elems(N, M); }
extern "C" uint64_t stb_vrowoff(unsigned n, unsigned M) { return stb_vrow_offset(n, M); }

// ------------------------------------------------------------------------------------------------
// the table of the stored log: {1/c, -log(1/c)} for c = 1 + (i + 1/2)/128, one copy per device

int stb_logtab(const double2 **out) {
  static std::mutex mu;
  static double2 *tab[64] = {nullptr};
  int dev = 0;
  HIPCHK(hipGetDevice(&dev));
  if (dev < 0 || dev >= 64) return stb_fail("device index %d out of range", dev);
  std::lock_guard<std::mutex> lk(mu);
  if (!tab[dev]) {
    double2 h[128];
    for (int i = 0; i < 128; i++) {
      const long double c = 1.0L + ((long double)i + 0.5L) / 128.0L;
      const double invc = (double)(1.0L / c);
      h[i].x = invc;
      h[i].y = (double)(-logl((long double)invc));
    }
    double2 *d = nullptr;
    HIPCHK(hipMalloc(&d, sizeof(h)));
    HIPCHK(hipMemcpy(d, h, sizeof(h), hipMemcpyHostToDevice));
    tab[dev] = d;
  }
  *out = tab[dev];
  return 0;
}
