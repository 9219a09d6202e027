// Optional in-library kernel timing: when enabled, every GEMM launch is
// bracketed by a pair of HIP events on its own stream.  Used by bench.py to
// report the dominant kernel's achieved FLOP/s from live measurements.  Off by
// default; while off nothing here touches the HIP runtime.
#include "../../include/ocean_hip.h"
#include "common.h"
#include <mutex>
#include <string>
#include <vector>
#include <map>
#include <cstdio>
#include <cstring>

namespace {
struct Rec { std::string name; double flops, bytes; hipEvent_t a, b; };
std::mutex g_mu;
bool g_on = false;
std::vector<Rec> g_recs;
}  // namespace

bool ocean_prof_on() { return g_on; }

void ocean_prof_begin(const char* name, double flops, double bytes, hipStream_t s, void** token) {
  *token = nullptr;
  if (!g_on) return;
  Rec* r = new Rec{name, flops, bytes, nullptr, nullptr};
  if (hipEventCreate(&r->a) != hipSuccess || hipEventCreate(&r->b) != hipSuccess) { delete r; return; }
  (void)hipEventRecord(r->a, s);
  *token = r;
}

void ocean_prof_end(void* token, hipStream_t s) {
  if (!token) return;
  Rec* r = static_cast<Rec*>(token);
  (void)hipEventRecord(r->b, s);
  std::lock_guard<std::mutex> lk(g_mu);
  g_recs.push_back(*r);
  delete r;
}

extern "C" {

int ocean_clear_last_error(void) { return (int)hipGetLastError(); }

int ocean_prof_enable(int on) {
  std::lock_guard<std::mutex> lk(g_mu);
  for (auto& r : g_recs) { (void)hipEventDestroy(r.a); (void)hipEventDestroy(r.b); }
  g_recs.clear();
  g_on = on != 0;
  return OCEAN_OK;
}

int ocean_prof_summary(char* buf, size_t cap) {
  if (!buf || cap == 0) return OCEAN_ERR_NULL;
  std::lock_guard<std::mutex> lk(g_mu);
  struct Agg { long n = 0; double ms = 0, flops = 0, bytes = 0; };
  std::map<std::string, Agg> agg;
  for (auto& r : g_recs) {
    if (hipEventSynchronize(r.b) != hipSuccess) continue;
    float ms = 0.f;
    if (hipEventElapsedTime(&ms, r.a, r.b) != hipSuccess) continue;
    Agg& a = agg[r.name];
    a.n++; a.ms += ms; a.flops += r.flops; a.bytes += r.bytes;
  }
  std::string out = "[";
  bool first = true;
  for (auto& kv : agg) {
    char line[512];
    snprintf(line, sizeof(line), "%s{\"kernel\":\"%s\",\"launches\":%ld,\"total_ms\":%.6f,\"flops\":%.6e,\"bytes\":%.6e}",
             first ? "" : ",", kv.first.c_str(), kv.second.n, kv.second.ms, kv.second.flops, kv.second.bytes);
    out += line;
    first = false;
  }
  out += "]";
  if (out.size() + 1 > cap) return OCEAN_ERR_WORKSPACE;
  memcpy(buf, out.c_str(), out.size() + 1);
  return OCEAN_OK;
}

}  // extern "C"
