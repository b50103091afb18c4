// C-ABI runtime glue: error reporting, library identity and an opt-in per-kernel event profiler
// (hipEvent pairs recorded on the launch stream around selected kernels, used by bench.py's roofline
// line).  No torch types anywhere in this library.
#include <hip/hip_runtime.h>
#include <stdarg.h>
#include <stdio.h>
#include <string.h>

#include <map>
#include <mutex>
#include <tuple>
#include <vector>

static thread_local char g_err[512] = "";

struct ProfRec {
  int kind;
  double work, bytes;
  hipEvent_t a, b;
};
static bool g_prof_on = false;
static std::vector<ProfRec> g_prof;
static std::vector<hipEvent_t> g_event_pool;

// ---- padded-raster lookup tables of the shifted-window kernels ------------------------------------
// The window kernels (conv_win.hip, conv_wgrad_win.hip) number pixels in a padded raster, G = (b (H+1) + h)(W+1) + w.
// Turning G back into a pixel index costs two divisions per lane -- ~25 VALU instructions per DMA instruction, which in the
// weight-gradient loop added up to 400 of 3250 cycles per chunk (round-3 probe).  The map depends on (B, H, W) only, so it
// is tabulated ONCE per geometry: tab[G + MPR_RASTER_MARGIN] = pixel index + 1, or 0 for a pad position / outside the
// raster.  Built on the host and copied synchronously (a few hundred KB .. 6.6 MB), kept for the life of the process.
#define MPR_RASTER_MARGIN 256
#define MPR_RASTER_TAIL 1024
static std::mutex g_raster_mu;
static std::map<std::tuple<int, int, int, int>, std::pair<uint32_t*, long long>> g_raster;

extern "C" const uint32_t* mpr_raster_table(int B, int H, int W, long long* entries) {
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess) return nullptr;
  std::lock_guard<std::mutex> lock(g_raster_mu);
  const auto key = std::make_tuple(dev, B, H, W);
  auto it = g_raster.find(key);
  if (it == g_raster.end()) {
    const long long img = (long long)(H + 1) * (W + 1), Gtot = (long long)B * img;
    const long long n = MPR_RASTER_MARGIN + Gtot + MPR_RASTER_TAIL;
    std::vector<uint32_t> host((size_t)n, 0u);
    uint32_t* q = host.data() + MPR_RASTER_MARGIN;
    for (int b = 0; b < B; ++b)
      for (int h = 0; h < H; ++h) {
        uint32_t* row = q + ((long long)b * (H + 1) + h) * (W + 1);
        const uint32_t pix1 = (uint32_t)(((long long)b * H + h) * W) + 1u;
        for (int w = 0; w < W; ++w) row[w] = pix1 + (uint32_t)w;
      }
    uint32_t* d = nullptr;
    if (hipMalloc((void**)&d, (size_t)n * 4) != hipSuccess) return nullptr;
    if (hipMemcpy(d, host.data(), (size_t)n * 4, hipMemcpyHostToDevice) != hipSuccess) {
      hipFree(d);
      return nullptr;
    }
    it = g_raster.emplace(key, std::make_pair(d, n)).first;
  }
  if (entries) *entries = it->second.second;
  return it->second.first;
}

extern "C" {

void mpr_set_error(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}

const char* mpr_last_error(void) { return g_err; }

int mpr_abi_version(void) { return 1; }

const char* mpr_target_arch(void) { return "gfx950"; }

// 0 when a gfx950 device is visible; fills `name` (>= 64 bytes) with the device's arch string.
int mpr_device_check(char* name, int name_len) {
  int n = 0;
  if (hipGetDeviceCount(&n) != hipSuccess || n == 0) {
    mpr_set_error("no HIP device visible");
    return 2;
  }
  hipDeviceProp_t prop;
  if (hipGetDeviceProperties(&prop, 0) != hipSuccess) {
    mpr_set_error("hipGetDeviceProperties failed");
    return 2;
  }
  if (name && name_len > 0) {
    strncpy(name, prop.gcnArchName, name_len - 1);
    name[name_len - 1] = 0;
  }
  if (strncmp(prop.gcnArchName, "gfx950", 6) != 0) {
    mpr_set_error("device 0 is %s, this library is built for gfx950 only", prop.gcnArchName);
    return 1;
  }
  return 0;
}

// ---- opt-in kernel profiler -------------------------------------------------------------------
// kinds: 0 conv_igemm forward, 1 conv_igemm dgrad, 2 conv_wgrad, 3 stem fwd, 4 stem wgrad
int mpr_prof_enable(int on) {
  g_prof_on = on != 0;
  return 0;
}

int mpr_prof_reset(void) {
  for (auto& r : g_prof) {
    g_event_pool.push_back(r.a);
    g_event_pool.push_back(r.b);
  }
  g_prof.clear();
  return 0;
}

static hipEvent_t prof_event() {
  if (!g_event_pool.empty()) {
    hipEvent_t e = g_event_pool.back();
    g_event_pool.pop_back();
    return e;
  }
  hipEvent_t e;
  hipEventCreate(&e);
  return e;
}

// internal: called by the launch wrappers (not part of the public header)
void* mpr_prof_begin(int kind, double work, void* stream) {
  if (!g_prof_on) return nullptr;
  ProfRec r;
  r.kind = kind;
  r.work = work;
  r.bytes = 0.0;
  r.a = prof_event();
  r.b = prof_event();
  hipEventRecord(r.a, (hipStream_t)stream);
  g_prof.push_back(r);
  return (void*)(uintptr_t)g_prof.size();
}

void mpr_prof_end(void* token, void* stream) {
  if (!token) return;
  ProfRec& r = g_prof[(size_t)(uintptr_t)token - 1];
  hipEventRecord(r.b, (hipStream_t)stream);
}

// internal: algorithmic HBM bytes (operands read once + results written once) of the launch behind `token`
void mpr_prof_bytes(void* token, double bytes) {
  if (token) g_prof[(size_t)(uintptr_t)token - 1].bytes = bytes;
}

int mpr_prof_collect_bytes(int kind, double* total_bytes) {
  double b = 0.0;
  for (auto& r : g_prof)
    if (kind < 0 || r.kind == kind) b += r.bytes;
  if (total_bytes) *total_bytes = b;
  return 0;
}

// Sums over all recorded launches of `kind` (-1: every kind).  Synchronises on the recorded events.
int mpr_prof_collect(int kind, double* total_ms, double* total_work, int* launches) {
  double ms = 0.0, work = 0.0;
  int n = 0;
  for (auto& r : g_prof) {
    if (kind >= 0 && r.kind != kind) continue;
    if (hipEventSynchronize(r.b) != hipSuccess) {
      mpr_set_error("mpr_prof_collect: hipEventSynchronize failed");
      return 2;
    }
    float t = 0.f;
    if (hipEventElapsedTime(&t, r.a, r.b) != hipSuccess) {
      mpr_set_error("mpr_prof_collect: hipEventElapsedTime failed");
      return 2;
    }
    ms += t;
    work += r.work;
    ++n;
  }
  if (total_ms) *total_ms = ms;
  if (total_work) *total_work = work;
  if (launches) *launches = n;
  return 0;
}

}  // extern "C"
