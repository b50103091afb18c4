// C-ABI runtime glue: error reporting and library identity.  No torch types anywhere in this library.
#include <hip/hip_runtime.h>
#include <stdarg.h>
#include <stdio.h>
#include <string.h>

static thread_local char g_err[512] = "";

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

}  // extern "C"
