// Error plumbing, version and device check for libdevit_hip.so.
#include <stdarg.h>
#include <stdio.h>
#include <string.h>

#include "devit_common.h"

static thread_local char g_err[512] = "";

void devit_set_error(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}

extern "C" int devit_version(void) { return DEVIT_ABI_VERSION; }
extern "C" const char* devit_last_error(void) { return g_err; }

extern "C" size_t devit_abi_struct_size(int which) {
  switch (which) {
    case 0: return sizeof(devit_epilogue);
    case 1: return sizeof(devit_operand);
    case 2: return sizeof(devit_block_weights);
    case 3: return sizeof(devit_block_wgrads);
    case 4: return sizeof(devit_block_acts);
    case 5: return sizeof(devit_block_bwd_io);
    case 6: return sizeof(devit_index_job);
    case 7: return sizeof(devit_wgrad_job);
    default: return 0;
  }
}

extern "C" int devit_check_device(int dev) {
  hipDeviceProp_t prop;
  hipError_t e = hipGetDeviceProperties(&prop, dev);
  DEVIT_CHECK(e == hipSuccess, DEVIT_ERR_DEVICE, "devit_check_device: %s", hipGetErrorString(e));
  DEVIT_CHECK(strncmp(prop.gcnArchName, "gfx950", 6) == 0, DEVIT_ERR_DEVICE,
              "devit_check_device: device %d is %s, this library is built for gfx950 only", dev, prop.gcnArchName);
  return DEVIT_OK;
}
