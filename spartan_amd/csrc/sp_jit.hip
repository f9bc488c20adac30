// Run-time specialisation ("third tier") of the fused-map / reduce kernels.
//
// The prebuilt StaticProg library covers the hot shapes; every other fused tree
// runs on the interpreter kernels, whose wave-uniform dispatch holds them at
// 25-70 % of the copy bandwidth (profiles/r01_notes.md).  For LARGE tiles this
// file closes the gap: it hands the very same hand-written evaluator source
// (sp_interp.hpp, map_kernel.hpp, reduce_impl.hpp) to hipRTC with the program's
// instruction stream spelled as one more `StaticProg` specialisation, so hipcc
// unrolls it exactly as it does for the prebuilt library.  Nothing is generated
// but that one struct; compiled code objects are cached per instruction stream.
// The reference's precedent is ParakeetExpr: a JIT-compiled local op installed
// when code generation succeeds, the interpreter otherwise
// (spartan/expr/operator/local.py:187-209, optimize.py:321-370).
//
// Failure of any step (no libhiprtc, compile error) is not an error: the caller
// falls back to the interpreter kernels, which are always correct.
#include <dlfcn.h>
#include <stdlib.h>

#include <condition_variable>
#include <deque>
#include <mutex>
#include <string>
#include <thread>
#include <unordered_map>
#include <vector>

#include <dirent.h>
#include <sys/stat.h>
#include <unistd.h>

#include "sp_jit.hpp"

namespace {

typedef struct _hiprtcProgram* rtcProgram;
struct Rtc {
  void* lib = nullptr;
  int (*create)(rtcProgram*, const char*, const char*, int, const char**, const char**) = nullptr;
  int (*compile)(rtcProgram, int, const char**) = nullptr;
  int (*addName)(rtcProgram, const char*) = nullptr;
  int (*lowered)(rtcProgram, const char*, const char**) = nullptr;
  int (*codeSize)(rtcProgram, size_t*) = nullptr;
  int (*code)(rtcProgram, char*) = nullptr;
  int (*logSize)(rtcProgram, size_t*) = nullptr;
  int (*log)(rtcProgram, char*) = nullptr;
  int (*destroy)(rtcProgram*) = nullptr;
  bool ok = false;
};

Rtc& rtc() {
  static Rtc r;
  static std::once_flag once;
  std::call_once(once, [] {
    const char* names[] = {"libhiprtc.so", "libhiprtc.so.7", "/opt/rocm/lib/libhiprtc.so"};
    for (const char* n : names) {
      r.lib = dlopen(n, RTLD_NOW | RTLD_LOCAL);
      if (r.lib) break;
    }
    if (!r.lib) return;
#define SP_SYM(field, name) *(void**)(&r.field) = dlsym(r.lib, name)
    SP_SYM(create, "hiprtcCreateProgram");
    SP_SYM(compile, "hiprtcCompileProgram");
    SP_SYM(addName, "hiprtcAddNameExpression");
    SP_SYM(lowered, "hiprtcGetLoweredName");
    SP_SYM(codeSize, "hiprtcGetCodeSize");
    SP_SYM(code, "hiprtcGetCode");
    SP_SYM(logSize, "hiprtcGetProgramLogSize");
    SP_SYM(log, "hiprtcGetProgramLog");
    SP_SYM(destroy, "hiprtcDestroyProgram");
#undef SP_SYM
    r.ok = r.create && r.compile && r.addName && r.lowered && r.codeSize && r.code && r.destroy;
  });
  return r;
}

bool verbose() {
  static int v = -1;
  if (v < 0) v = getenv("SP_JIT_VERBOSE") ? 1 : 0;
  return v != 0;
}

// directory of this shared library == spartan_amd/csrc (the headers ship next to it)
std::string source_dir() {
  Dl_info info;
  if (dladdr((const void*)&sp_jit_enabled, &info) && info.dli_fname) {
    std::string path(info.dli_fname);
    size_t slash = path.rfind('/');
    if (slash != std::string::npos) return path.substr(0, slash);
  }
  return ".";
}

std::string program_struct(const sp_program* p) {
  std::string s = "template <> struct StaticProg<1000> {\n  static constexpr bool kStatic = true;\n";
  s += "  static constexpr int N = " + std::to_string(p->n_instr) + ";\n";
  s += "  static constexpr int NIN = " + std::to_string(p->n_inputs) + ";\n";
  s += "  static constexpr int RESULT = " + std::to_string(p->result_reg) + ";\n";
  s += "  static __host__ __device__ constexpr sp_instr at(int pc) {\n    switch (pc) {\n";
  for (int i = 0; i < p->n_instr; ++i) {
    const sp_instr& I = p->instr[i];
    s += "      case " + std::to_string(i) + ": return sp_instr{" + std::to_string(I.op) + ", " +
         std::to_string(I.dst) + ", " + std::to_string(I.a) + ", " + std::to_string(I.b) + ", " +
         std::to_string(I.c) + ", 0, 0, 0};\n";
  }
  s += "      default: return sp_instr{0, 0, 0, 0, 0, 0, 0, 0};\n    }\n  }\n";
  s += "  static __host__ __device__ constexpr int in_dtype(int j) {\n    switch (j) {\n";
  for (int j = 0; j < p->n_inputs; ++j)
    s += "      case " + std::to_string(j) + ": return " + std::to_string(p->in_dtype[j]) + ";\n";
  s += "      default: return 0;\n    }\n  }\n};\n";
  return s;
}

std::string program_key(const char* header, const char* expr, const sp_program* p) {
  std::string k = std::string(header) + "|" + expr + "|" + std::to_string(p->cls) + "|" +
                  std::to_string(p->n_inputs) + "|" + std::to_string(p->result_reg) + "|";
  for (int j = 0; j < p->n_inputs; ++j) k += std::to_string(p->in_dtype[j]) + ",";
  k += "|";
  for (int i = 0; i < p->n_instr; ++i) {
    const sp_instr& I = p->instr[i];
    char buf[48];
    snprintf(buf, sizeof(buf), "%u.%u.%u.%u.%u;", I.op, I.dst, I.a, I.b, I.c);
    k += buf;
  }
  return k;
}

// ---- code objects kept across processes (opt-in: SPARTAN_JIT_CACHE=<directory>) -------------------------------
// One file per (kernel sources, header, template expression, program): the lowered kernel name and the code object
// hipRTC produced.  The sources' id is a hash of the CONTENTS of the headers the kernels are instantiated from, so
// changed sources never load code compiled from older ones, and a copy of the tree elsewhere keeps its files valid.
uint64_t fnv1a(const std::string& s, uint64_t h = 1469598103934665603ull) {
  for (unsigned char c : s) {
    h ^= c;
    h *= 1099511628211ull;
  }
  return h;
}

// Where specialised code objects persist across processes.  On by default: $SPARTAN_JIT_CACHE, else
// $XDG_CACHE_HOME/spartan_amd/jit, else ~/.cache/spartan_amd/jit (created on first use); SPARTAN_JIT_CACHE=off (or
// an empty value) keeps them in memory only.  A second process that evaluates the same fused expression then
// starts on the specialised kernel instead of spending its first launches on the interpreter tier.
std::string cache_dir() {
  static std::string dir;
  static bool resolved = false;
  if (resolved) return dir;
  resolved = true;
  const char* e = getenv("SPARTAN_JIT_CACHE");
  if (e) {
    dir = (*e && strcmp(e, "off") != 0 && strcmp(e, "0") != 0) ? std::string(e) : std::string();
  } else {
    const char* x = getenv("XDG_CACHE_HOME");
    const char* h = getenv("HOME");
    if (x && *x) dir = std::string(x) + "/spartan_amd/jit";
    else if (h && *h) dir = std::string(h) + "/.cache/spartan_amd/jit";
  }
  if (!dir.empty()) {
    // mkdir -p (errors are not fatal: an unwritable cache only means every process compiles for itself)
    for (size_t i = 1; i <= dir.size(); ++i)
      if (i == dir.size() || dir[i] == '/') mkdir(dir.substr(0, i).c_str(), 0755);
  }
  return dir;
}

// identity of a source file by CONTENT (a copy of the tree on another machine keeps it; size + mtime would not)
std::string content_id(const std::string& path) {
  FILE* f = fopen(path.c_str(), "rb");
  if (!f) return "?";
  uint64_t h = 1469598103934665603ull;
  char buf[65536];
  size_t n;
  while ((n = fread(buf, 1, sizeof buf, f)) > 0)
    for (size_t i = 0; i < n; ++i) {
      h ^= (unsigned char)buf[i];
      h *= 1099511628211ull;
    }
  fclose(f);
  char out[24];
  snprintf(out, sizeof out, "%016llx", (unsigned long long)h);
  return out;
}

// the kernels are instantiated from these headers only
const std::string& sources_id() {
  static std::string id;
  static std::once_flag once;
  std::call_once(once, [] {
    const std::string dir = source_dir();
    for (const char* f : {"sp_common.hpp", "sp_interp.hpp", "map_kernel.hpp", "reduce_impl.hpp", "sp_jit.hip",
                          "../../include/spartan_hip.h"})
      id += content_id(dir + "/" + f) + "|";
  });
  return id;
}

// Code objects that travel with the tree: <csrc>/jit_seed, filled by __graft_entry__.build() (spartan_amd/jit_seed.py
// drives the real launch path in seed mode on a machine without a GPU) for the fused programs the workloads are
// known to force, so that on a fresh machine their FIRST launch already runs specialised.
std::string seed_dir() { return source_dir() + "/jit_seed"; }
std::string g_seed_target;     // non-empty: seed mode (sp_jit_seed_begin)

std::string cache_file(const char* header, const char* expr, const sp_program* p);

hipFunction_t cache_load(const std::string& path, std::string* symbol = nullptr) {
  FILE* f = fopen(path.c_str(), "rb");
  if (!f) return nullptr;
  hipFunction_t fn = nullptr;
  uint32_t name_len = 0;
  uint64_t code_len = 0;
  if (fread(&name_len, 4, 1, f) == 1 && name_len < 4096 && fread(&code_len, 8, 1, f) == 1 && code_len < (1ull << 30)) {
    std::string name(name_len, 0);
    std::vector<char> code(code_len);
    if (fread(&name[0], 1, name_len, f) == name_len && fread(code.data(), 1, code_len, f) == code_len) {
      hipModule_t mod = nullptr;
      if (hipModuleLoadData(&mod, code.data()) == hipSuccess && hipModuleGetFunction(&fn, mod, name.c_str()) != hipSuccess)
        fn = nullptr;
      (void)hipGetLastError();
      if (fn && symbol) *symbol = name;
    }
  }
  fclose(f);
  return fn;
}

void cache_store(const std::string& path, const char* lowered, const std::vector<char>& code) {
  const std::string tmp = path + ".tmp" + std::to_string((long long)getpid());
  FILE* f = fopen(tmp.c_str(), "wb");
  if (!f) return;
  const uint32_t name_len = (uint32_t)strlen(lowered);
  const uint64_t code_len = code.size();
  const bool ok = fwrite(&name_len, 4, 1, f) == 1 && fwrite(&code_len, 8, 1, f) == 1 &&
                  fwrite(lowered, 1, name_len, f) == name_len && fwrite(code.data(), 1, code_len, f) == code_len;
  fclose(f);
  if (!ok || rename(tmp.c_str(), path.c_str()) != 0) remove(tmp.c_str());   // (rename: readers never see half a file)
}

std::mutex g_mu;
std::unordered_map<std::string, hipFunction_t> g_cache;   // value NULL = pending, or tried and failed
std::unordered_map<std::string, hipFunction_t> g_preloaded;   // "<device>/<file>.spco" -> function (sp_jit_preload)

// Compiles run on ONE background thread: a launch that finds its program not yet
// specialised enqueues the job and carries on with the interpreter kernel, so the
// ~0.4 s hipRTC compile is never on the launch path; later launches of the same
// program pick the specialised kernel up once it is ready.  (Tiers are bit-identical,
// so the switch is invisible in the results.)  SP_JIT_SYNC=1 compiles in the caller.
struct Job {
  std::string key, header, expr;
  sp_program prog;
  int device;
};
std::deque<Job> g_jobs;
std::condition_variable g_cv, g_idle_cv;
std::thread g_worker;
bool g_worker_started = false, g_stop = false;
int g_inflight = 0;

hipFunction_t compile(const char* header, const char* expr, const sp_program* p, bool load = true, int* compiled = nullptr) {
  Rtc& r = rtc();
  if (!r.ok) return nullptr;
  const std::string dir = source_dir();
  const std::string file = cache_file(header, expr, p);
  const std::string disk = load && !cache_dir().empty() ? cache_dir() + file : std::string();
  if (load) {
    for (const std::string& where : {disk, seed_dir() + file}) {
      if (where.empty()) continue;
      hipFunction_t cached = cache_load(where);
      if (cached) {
        if (verbose()) fprintf(stderr, "[spartan_hip jit] loaded %s from %s\n", expr, where.c_str());
        return cached;
      }
    }
  }
  std::string src = "#include \"" + std::string(header) + "\"\n" + program_struct(p);
  rtcProgram prog = nullptr;
  if (r.create(&prog, src.c_str(), "sp_jit_program.hip", 0, nullptr, nullptr) != 0) return nullptr;
  r.addName(prog, expr);
  // (hipRTC normally serves <hip/hip_runtime.h> from its built-in copy; under rocprofv3 it does not:
  //  the ROCm include directory -- after the shims, which must win for the C headers -- is the retry)
  const char* rocm = getenv("ROCM_PATH");
  const std::string inc1 = "-I" + dir + "/rtc_shim", inc2 = "-I" + dir,
                    inc3 = std::string("-I") + (rocm && *rocm ? rocm : "/opt/rocm") + "/include";
  const char* opts[] = {"--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off", inc1.c_str(), inc2.c_str(),
                        inc3.c_str()};
  int rc = r.compile(prog, 6, opts);            // built-in HIP headers: the fast, usual case
  if (rc != 0) {
    // retry with the installed ROCm headers, on a fresh program object (a failed compile leaves its
    // partial outputs in the old one: "duplicate symbol" at link)
    rtcProgram again = nullptr;
    if (r.create(&again, src.c_str(), "sp_jit_program.hip", 0, nullptr, nullptr) == 0) {
      r.destroy(&prog);
      prog = again;
      r.addName(prog, expr);
      rc = r.compile(prog, 7, opts);
    }
  }
  hipFunction_t fn = nullptr;
  if (rc != 0) {
    if (verbose() && r.logSize && r.log) {
      size_t n = 0;
      r.logSize(prog, &n);
      std::vector<char> log(n + 1, 0);
      r.log(prog, log.data());
      fprintf(stderr, "[spartan_hip jit] compile failed for %s:\n%s\n", expr, log.data());
    }
  } else if (!load) {
    if (compiled) *compiled = 1;
    if (!g_seed_target.empty()) {        // seed mode: keep the code object, do not load it (there may be no device)
      const char* lowered = nullptr;
      size_t n = 0;
      if (r.lowered(prog, expr, &lowered) == 0 && lowered && r.codeSize(prog, &n) == 0 && n) {
        std::vector<char> code(n);
        if (r.code(prog, code.data()) == 0) cache_store(g_seed_target + file, lowered, code);
      }
    }
  } else {
    const char* lowered = nullptr;
    size_t n = 0;
    if (r.lowered(prog, expr, &lowered) == 0 && lowered && r.codeSize(prog, &n) == 0 && n) {
      std::vector<char> code(n);
      if (r.code(prog, code.data()) == 0) {
        hipModule_t mod = nullptr;
        if (hipModuleLoadData(&mod, code.data()) == hipSuccess) {
          if (hipModuleGetFunction(&fn, mod, lowered) != hipSuccess) fn = nullptr;
        }
        (void)hipGetLastError();
        if (fn && !disk.empty()) cache_store(disk, lowered, code);
      }
    }
    if (verbose()) fprintf(stderr, "[spartan_hip jit] %s %s (%d instrs)\n", fn ? "compiled" : "load failed", expr, p->n_instr);
  }
  r.destroy(&prog);
  return fn;
}

std::string cache_file(const char* header, const char* expr, const sp_program* p) {
  char name[64];
  snprintf(name, sizeof(name), "/%016llx.spco", (unsigned long long)fnv1a(program_key(header, expr, p), fnv1a(sources_id())));
  return name;
}

}  // namespace

static int g_enabled = -1;
static long long g_min_elems = -1;
static int g_seeded = 0;

int sp_jit_enabled() {
  if (g_enabled < 0) g_enabled = getenv("SP_NO_JIT") == nullptr ? 1 : 0;
  return g_enabled && rtc().ok;
}

int64_t sp_jit_min_elems() {
  if (g_min_elems < 0) {
    const char* e = getenv("SP_JIT_MIN_ELEMS");
    g_min_elems = e ? atoll(e) : (1LL << 22);
  }
  return g_min_elems;
}

extern "C" int sp_jit_configure(int enabled, long long min_elems) {
  if (enabled >= 0) g_enabled = enabled ? 1 : 0;
  if (min_elems >= 0) g_min_elems = min_elems;
  return sp_jit_enabled();
}

static void worker_main() {
  for (;;) {
    Job job;
    {
      std::unique_lock<std::mutex> lock(g_mu);
      g_cv.wait(lock, [] { return g_stop || !g_jobs.empty(); });
      if (g_stop) return;
      job = std::move(g_jobs.front());
      g_jobs.pop_front();
    }
    hipFunction_t fn = nullptr;
    if (hipSetDevice(job.device) == hipSuccess) fn = compile(job.header.c_str(), job.expr.c_str(), &job.prog);
    {
      std::lock_guard<std::mutex> lock(g_mu);
      if (!g_stop) g_cache[job.key] = fn;
      --g_inflight;
    }
    g_idle_cv.notify_all();
  }
}

static void stop_worker() {
  {
    std::lock_guard<std::mutex> lock(g_mu);
    g_stop = true;
  }
  g_cv.notify_all();
  if (g_worker.joinable()) g_worker.join();
}

static int sync_mode() {
  static int v = -1;
  if (v < 0) v = getenv("SP_JIT_SYNC") ? 1 : 0;
  return v;
}

void* sp_jit_get(const char* header, const char* template_expr, const sp_program* p) {
  if (!sp_jit_enabled()) return nullptr;
  if (!g_seed_target.empty()) {         // seed mode: compile now, store, and let the caller go on (to nowhere)
    std::lock_guard<std::mutex> lock(g_mu);
    const std::string key = "seed|" + program_key(header, template_expr, p);
    if (g_cache.find(key) == g_cache.end()) {
      int ok = 0;
      compile(header, template_expr, p, false, &ok);
      g_cache.emplace(key, nullptr);
      g_seeded += ok;
    }
    return nullptr;
  }
  int device = 0;
  if (hipGetDevice(&device) != hipSuccess) return nullptr;
  const std::string key = std::to_string(device) + "|" + program_key(header, template_expr, p);
  std::unique_lock<std::mutex> lock(g_mu);
  auto it = g_cache.find(key);
  if (it != g_cache.end()) return (void*)it->second;
  if (sync_mode()) {
    hipFunction_t fn = compile(header, template_expr, p);
    g_cache.emplace(key, fn);
    return (void*)fn;
  }
  if (g_stop) return nullptr;
  {
    // a code object already on disk (the user's cache, or the seeds that travel with the tree) is loaded right
    // here -- milliseconds, not a compile -- so that such a program runs specialised from its FIRST launch;
    // sp_jit_preload (called in the background when the backend comes up) may have loaded it already
    const std::string file = cache_file(header, template_expr, p);
    auto pre = g_preloaded.find(std::to_string(device) + file);
    if (pre != g_preloaded.end()) {
      g_cache.emplace(key, pre->second);
      return (void*)pre->second;
    }
    for (const std::string& where : {cache_dir().empty() ? std::string() : cache_dir() + file, seed_dir() + file}) {
      if (where.empty()) continue;
      hipFunction_t fn = cache_load(where);
      if (fn) {
        if (verbose()) fprintf(stderr, "[spartan_hip jit] loaded %s from %s\n", template_expr, where.c_str());
        g_cache.emplace(key, fn);
        return (void*)fn;
      }
    }
  }
  g_cache.emplace(key, nullptr);   // pending: callers use the interpreter meanwhile
  g_jobs.push_back(Job{key, header, template_expr, *p, device});
  ++g_inflight;
  if (!g_worker_started) {
    g_worker_started = true;
    // libhiprtc (and the compiler library it brings) is loaded HERE, before the exit handler is registered: exit
    // handlers run in reverse order, so the compiler's own teardown comes after stop_worker has joined a compile
    // that may be in flight (loaded lazily on the worker thread it was torn down under a running compile: a
    // short-lived process could hang at exit)
    (void)rtc();
    g_worker = std::thread(worker_main);
    atexit(stop_worker);   // registered after the HIP runtime came up => runs before its teardown
  }
  lock.unlock();
  g_cv.notify_one();
  return nullptr;
}

// Loads every code object of the seed directory and of SPARTAN_JIT_CACHE onto `device` (about a millisecond each),
// so that the first launch of a seeded program does not wait for the file and hipModuleLoadData.  Meant to run on a
// background thread while the host builds its first expressions; returns the number of functions loaded.
extern "C" int sp_jit_preload(int device) {
  if (!sp_jit_enabled() || !g_seed_target.empty()) return 0;
  if (hipSetDevice(device) != hipSuccess) {
    (void)hipGetLastError();
    return 0;
  }
  // (the identity of the sources -- a hash of six files' contents -- and the cache directory are worked out here, on
  // this thread: left to the first sp_jit_get they were ~0.2 ms of the first launch of a seeded program)
  (void)sources_id();
  (void)cache_dir();
  int loaded = 0;
  // The first launch of a function pays for its set-up on the device (hundreds of microseconds measured for a
  // preloaded map kernel: 0.64 ms against 0.012 ms for every later launch).  Map kernels share one signature and do
  // nothing for nvec = 0, so each is launched once, empty, on a stream of this thread: the first real launch of a
  // seeded program is then an ordinary launch.  The reduce kernels do nothing for an empty outer range (O = 0) and
  // get the same treatment (round 5: the first launch of a seeded fused reduction on the 2 GiB tile took 0.75 ms for
  // a 0.33 ms kernel).
  hipStream_t warm = nullptr;
  if (hipStreamCreateWithFlags(&warm, hipStreamNonBlocking) != hipSuccess) warm = nullptr;
  (void)hipGetLastError();
  for (const std::string& dir : {cache_dir(), seed_dir()}) {
    if (dir.empty()) continue;
    DIR* d = opendir(dir.c_str());
    if (!d) continue;
    std::vector<std::string> files;
    while (struct dirent* e = readdir(d)) {
      const std::string f = e->d_name;
      if (f.size() > 5 && f.compare(f.size() - 5, 5, ".spco") == 0) files.push_back("/" + f);
    }
    closedir(d);
    for (const std::string& f : files) {
      const std::string key = std::to_string(device) + f;
      {
        std::lock_guard<std::mutex> lock(g_mu);
        if (g_stop) return loaded;
        if (g_preloaded.count(key)) continue;
      }
      std::string symbol;
      hipFunction_t fn = cache_load(dir + f, &symbol);
      if (!fn) continue;
      if (warm && symbol.find("sp_map_kernel") != std::string::npos) {
        sp_program p0;
        sp_inputs in0;
        memset(&p0, 0, sizeof p0);
        memset(&in0, 0, sizeof in0);
        p0.ndim = 1;
        p0.shape[0] = 1;
        void* out0 = nullptr;
        int64_t start0 = 0, nvec0 = 0;
        void* args[] = {&p0, &in0, &out0, &start0, &nvec0};
        (void)hipModuleLaunchKernel(fn, 1, 1, 1, SP_BLOCK, 1, 1, 0, warm, args, nullptr);
        (void)hipGetLastError();
      }
      if (warm && symbol.find("sp_reduce_") != std::string::npos) {
        // (p, in, op, O, A[, I], [chunk, nsplit,] RedOut[, c0]): see reduce_impl.hpp; RedOut is 56 bytes of pointers
        // and integers (reduce.hip asserts the size), all zero here
        sp_program p0;
        sp_inputs in0;
        memset(&p0, 0, sizeof p0);
        memset(&in0, 0, sizeof in0);
        p0.ndim = 1;
        p0.shape[0] = 1;
        int op0 = 0, nsplit0 = 1;
        int64_t zero = 0, one = 1;
        unsigned char ro0[56];
        memset(ro0, 0, sizeof ro0);
        const bool cols = symbol.find("sp_reduce_cols_kernel") != std::string::npos;
        const bool wave = symbol.find("sp_reduce_rows_wave_kernel") != std::string::npos;
        const bool rows = !cols && !wave && symbol.find("sp_reduce_rows_kernel") != std::string::npos;
        if (cols) {
          void* args[] = {&p0, &in0, &op0, &zero, &one, &one, &one, &nsplit0, ro0, &zero};
          (void)hipModuleLaunchKernel(fn, 1, 1, 1, SP_BLOCK, 1, 1, 0, warm, args, nullptr);
        } else if (rows) {
          void* args[] = {&p0, &in0, &op0, &zero, &one, &one, &nsplit0, ro0};
          (void)hipModuleLaunchKernel(fn, 1, 1, 1, SP_BLOCK, 1, 1, 0, warm, args, nullptr);
        } else if (wave) {
          void* args[] = {&p0, &in0, &op0, &zero, &one, ro0};
          (void)hipModuleLaunchKernel(fn, 1, 1, 1, SP_BLOCK, 1, 1, 0, warm, args, nullptr);
        }
        (void)hipGetLastError();
      }
      std::lock_guard<std::mutex> lock(g_mu);
      g_preloaded.emplace(key, fn);
      ++loaded;
    }
  }
  if (warm) {
    (void)hipStreamSynchronize(warm);
    (void)hipStreamDestroy(warm);
    (void)hipGetLastError();
  }
  if (verbose()) fprintf(stderr, "[spartan_hip jit] preloaded %d code objects on device %d\n", loaded, device);
  return loaded;
}

// Stops the compile thread (an in-flight compile finishes first); launches after this use what is already compiled,
// loaded or interpreted.  The Python host calls it from its own atexit, ahead of every C exit handler.
extern "C" void sp_jit_shutdown(void) { stop_worker(); }

// Block until every queued specialisation has been compiled (tests, benchmarks).
extern "C" void sp_jit_wait(void) {
  std::unique_lock<std::mutex> lock(g_mu);
  g_idle_cv.wait(lock, [] { return g_inflight == 0 || g_stop; });
}

int sp_jit_launch(void* fn, dim3 grid, dim3 block, void** args, hipStream_t st) {
  SP_HIP(hipModuleLaunchKernel((hipFunction_t)fn, grid.x, grid.y, grid.z, block.x, block.y, block.z, 0, st, args,
                               nullptr));
  return 0;
}

// Test / diagnostics hook: does `template_expr` compile for `p` (no device needed)?
// Returns 1 when hipRTC produced a code object, 0 otherwise.
extern "C" int sp_jit_compile_check(const char* header, const char* template_expr, const sp_program* p) {
  int ok = 0;
  compile(header, template_expr, p, false, &ok);
  return ok;
}

// Seed mode (spartan_amd/jit_seed.py, called by __graft_entry__.build()): between begin and end every
// specialisation a launch asks for is compiled at once and written to `dir` (default: <csrc>/jit_seed) instead of
// being loaded -- no device is needed; the launches themselves fail and are ignored by the caller.  Returns the
// number of code objects written.
extern "C" int sp_jit_seed_begin(const char* dir) {
  std::lock_guard<std::mutex> lock(g_mu);
  g_seed_target = dir && *dir ? std::string(dir) : seed_dir();
  for (size_t i = 1; i <= g_seed_target.size(); ++i)
    if (i == g_seed_target.size() || g_seed_target[i] == '/') mkdir(g_seed_target.substr(0, i).c_str(), 0755);
  g_seeded = 0;
  return rtc().ok ? 1 : 0;
}

extern "C" int sp_jit_seed_end(void) {
  std::lock_guard<std::mutex> lock(g_mu);
  g_seed_target.clear();
  return g_seeded;
}

// Test / diagnostics hook: number of run-time specialised kernels compiled so far.
extern "C" int sp_jit_compiled_count(void) {
  std::lock_guard<std::mutex> lock(g_mu);
  int n = 0;
  for (auto& kv : g_cache)
    if (kv.second) ++n;
  return n;
}
