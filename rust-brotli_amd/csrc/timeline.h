// timeline.h -- BROTLI_MI355X_TIMELINE=1: host-side timestamps (ms since the call started) of the phases of one
// EncodeStream call, printed to stderr when the call returns.  Costs one predictable branch when off.
#ifndef BROTLI_MI355X_TIMELINE_H_
#define BROTLI_MI355X_TIMELINE_H_
#include <stdio.h>
#include <stdlib.h>

#include <chrono>
#include <string>

#include "device_api.h"

namespace brotli_mi355x {

struct Timeline {
  bool on = getenv("BROTLI_MI355X_TIMELINE") != nullptr;
  std::chrono::steady_clock::time_point t0, last_end;
  std::string text;
  void begin() {
    if (!on) return;
    t0 = std::chrono::steady_clock::now();
    char buf[64];
    snprintf(buf, sizeof(buf), "(%.3f ms since the last call returned)", std::chrono::duration<double, std::milli>(t0 - last_end).count());
    text = buf;
  }
  void stamp(const char* what) {
    if (!on) return;
    char buf[96];
    snprintf(buf, sizeof(buf), " %s %.3f", what, std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count());
    text += buf;
  }
  void end() {
    if (!on) return;
    stamp("returns");
    double pc[4];
    dev_pool_counters(pc, true);
    fprintf(stderr, "timeline: %s | hipMalloc %.0f calls %.3f ms, hipFree %.0f calls %.3f ms\n", text.c_str(), pc[0], pc[1], pc[2], pc[3]);
    last_end = std::chrono::steady_clock::now();
  }
};
inline Timeline& timeline() {
  static thread_local Timeline t;
  return t;
}

}  // namespace brotli_mi355x
#endif
