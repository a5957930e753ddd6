// Library-wide pieces of the C ABI: version, last-error string, device probe.
#include "common.h"

static thread_local char gq_err_[512] = "";

extern "C" {

void gq_set_error_(const char* msg) {
  strncpy(gq_err_, msg, sizeof(gq_err_) - 1);
  gq_err_[sizeof(gq_err_) - 1] = 0;
}

const char* gq_last_error(void) { return gq_err_; }

int gq_version(void) { return 100; }  // 0.1.0

// 0 if a gfx950-class device is usable from this process, else an error code with gq_last_error() set
int gq_device_check(int device, char* arch_out, int arch_len) {
  int n = 0;
  GQ_CHECK_HIP(hipGetDeviceCount(&n));
  GQ_REQUIRE(device >= 0 && device < n, "device_check: device %d not present (%d devices)", device, n);
  hipDeviceProp_t p;
  GQ_CHECK_HIP(hipGetDeviceProperties(&p, device));
  if (arch_out && arch_len > 0) {
    strncpy(arch_out, p.gcnArchName, arch_len - 1);
    arch_out[arch_len - 1] = 0;
  }
  GQ_REQUIRE(strncmp(p.gcnArchName, "gfx950", 6) == 0, "device_check: %s is not gfx950 (this library ships gfx950 code only)",
             p.gcnArchName);
  return GQ_OK;
}

// ---- gqTimer: a pair of HIP events that a launch fills with the kernel's own start / stop timestamps ----------
int gq_timer_create(void** out) {
  GQ_REQUIRE(out, "timer_create: null");
  hipEvent_t* p = new hipEvent_t[2];
  GQ_CHECK_HIP(hipEventCreate(&p[0]));
  GQ_CHECK_HIP(hipEventCreate(&p[1]));
  *out = p;
  return GQ_OK;
}

int gq_timer_elapsed_ms(void* timer, float* ms) {  // synchronises on the stop event
  GQ_REQUIRE(timer && ms, "timer_elapsed_ms: null");
  hipEvent_t* p = (hipEvent_t*)timer;
  GQ_CHECK_HIP(hipEventSynchronize(p[1]));
  GQ_CHECK_HIP(hipEventElapsedTime(ms, p[0], p[1]));
  return GQ_OK;
}

int gq_timer_destroy(void* timer) {
  if (!timer) return GQ_OK;
  hipEvent_t* p = (hipEvent_t*)timer;
  (void)hipEventDestroy(p[0]);
  (void)hipEventDestroy(p[1]);
  delete[] p;
  return GQ_OK;
}

}  // extern "C"
