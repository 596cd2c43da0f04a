// libpaule_hip.so -- the C-ABI of include/paule_hip.h as a LOADER: no device code and no link-time dependency on a HIP runtime.
//
// Why.  A process must run on ONE HIP runtime: two copies of libamdhip64 mapped side by side each keep their
// own device state, and the one that initialises second finds "no ROCm-capable device".  The library used to be linked against
// /opt/rocm's libamdhip64.so.7 (NEEDED + RUNPATH); a host that already carries another copy -- PyTorch's wheel ships its own,
// under another soname -- then got two, and which one won depended on the ORDER in which the libraries were loaded (seen on the GPU
// box: build() then smoke() in one Python process).  Now:
//
//   libpaule_hip.so       this file: every pl_* entry point, forwarded; plain C++, links against libdl only
//   libpaule_hip_core.so  the kernels and the planner (hipcc objects), linked WITHOUT a HIP runtime: its hip* symbols are undefined
//
// On the first call the loader (1) looks for a HIP runtime that is ALREADY mapped into the process (dl_iterate_phdr: any object whose
// file name starts with libamdhip64.so) and promotes it to the global symbol scope (dlopen RTLD_NOLOAD | RTLD_GLOBAL) -- whatever the
// host brought, in whatever order; (2) only if there is none, loads one: $PAULE_HIP_RUNTIME, else libamdhip64.so by the normal search
// path, else ${ROCM_PATH:-/opt/rocm}/lib/libamdhip64.so; (3) loads libpaule_hip_core.so from its own directory ($PAULE_HIP_CORE
// overrides: diagnostic builds), whose undefined hip* symbols bind to that one runtime; (4) resolves the entry points.  If MORE than
// one runtime is mapped (a host that linked /opt/rocm's AND imported torch), the first one mapped is used, and a pl_create that fails
// names the situation in pl_last_error() ("two HIP runtimes are mapped into this process: ...").  A load failure makes every entry point return
// PL_ERR_HIP with the reason in pl_last_error(); nothing aborts.
#include <dlfcn.h>
#include <link.h>
#include <stdint.h>

#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <string>
#include <vector>

#include "../../include/paule_hip.h"

namespace {

struct Loader {
    std::once_flag once;
    void* core = nullptr;
    std::string error;            // why the core is not there (empty: loaded)
    std::string runtime_path;
#define PL_FWD(ret, name, typed, args) ret(*p_##name) typed = nullptr;
#define PL_FWD_CUSTOM PL_FWD
#include "shim_table.inc"
#undef PL_FWD_CUSTOM
#undef PL_FWD
};
Loader g;
thread_local std::string t_msg;   // what pl_last_error returns when the loader itself has something to say

std::string dl_err(const std::string& what) {   // dlerror() clears the message it returns: read it ONCE
    const char* e = dlerror();
    return e ? std::string(e) : what + ": unknown dlopen error";
}

int collect_runtimes(struct dl_phdr_info* info, size_t, void* data) {
    const char* name = info->dlpi_name;
    if (!name || !*name) return 0;
    const char* base = std::strrchr(name, '/');
    base = base ? base + 1 : name;
    if (std::strncmp(base, "libamdhip64.so", 14) == 0) static_cast<std::vector<std::string>*>(data)->push_back(name);
    return 0;
}

std::string own_directory() {
    Dl_info di{};
    if (dladdr(reinterpret_cast<void*>(&collect_runtimes), &di) && di.dli_fname) {
        std::string p(di.dli_fname);
        const size_t s = p.rfind('/');
        return s == std::string::npos ? std::string(".") : p.substr(0, s);
    }
    return ".";
}

void load() {
    // (1) a runtime the process already has
    std::vector<std::string> mapped;
    dl_iterate_phdr(collect_runtimes, &mapped);
    void* rt = nullptr;
    if (!mapped.empty()) {
        rt = dlopen(mapped[0].c_str(), RTLD_NOW | RTLD_NOLOAD | RTLD_GLOBAL);   // promote it: the core's undefined symbols bind to THIS copy
        if (!rt) {
            g.error = "a HIP runtime is mapped (" + mapped[0] + ") but could not be promoted to the global scope: " + dl_err(mapped[0]);
            return;
        }
        g.runtime_path = mapped[0];
    } else {
        // (2) none yet: bring one
        std::vector<std::string> tries;
        if (const char* e = std::getenv("PAULE_HIP_RUNTIME")) tries.push_back(e);
        tries.push_back("libamdhip64.so");
        tries.push_back("libamdhip64.so.7");
        const char* rocm = std::getenv("ROCM_PATH");
        tries.push_back(std::string(rocm && *rocm ? rocm : "/opt/rocm") + "/lib/libamdhip64.so");
        std::string errs;
        for (const std::string& t : tries) {
            rt = dlopen(t.c_str(), RTLD_NOW | RTLD_GLOBAL);
            if (rt) { g.runtime_path = t; break; }
            errs += std::string(errs.empty() ? "" : "; ") + dl_err(t);
        }
        if (!rt) {
            g.error = "no HIP runtime is mapped into this process and none could be loaded (" + errs + "); set PAULE_HIP_RUNTIME or ROCM_PATH";
            return;
        }
    }
    // (3) the kernels + planner
    std::string core_path;
    if (const char* e = std::getenv("PAULE_HIP_CORE")) core_path = e;
    else core_path = own_directory() + "/libpaule_hip_core.so";
    g.core = dlopen(core_path.c_str(), RTLD_NOW | RTLD_LOCAL);
    if (!g.core) {
        g.error = "cannot load " + core_path + ": " + dl_err(core_path) + " (HIP runtime in use: " + g.runtime_path + ")";
        return;
    }
    // (4) entry points
#define PL_FWD(ret, name, typed, args)                                                             \
    g.p_##name = reinterpret_cast<ret(*) typed>(dlsym(g.core, #name));                             \
    if (!g.p_##name && g.error.empty()) g.error = core_path + " does not export " #name " (stale build?)";
#define PL_FWD_CUSTOM PL_FWD
#include "shim_table.inc"
#undef PL_FWD_CUSTOM
#undef PL_FWD
    if (!g.error.empty()) { g.core = nullptr; return; }
    // (5) the runtime the kernels are now bound to against the one they were compiled for: another MAJOR release is refused
    using rt_ver_fn = int (*)(int*);
    rt_ver_fn rt_ver = reinterpret_cast<rt_ver_fn>(dlsym(rt, "hipRuntimeGetVersion"));
    int have = 0;
    if (rt_ver && rt_ver(&have) == 0 && have > 0 && g.p_pl_hip_version_built) {
        const int built = g.p_pl_hip_version_built();
        if (have / 10000000 != built / 10000000) {
            g.error = "the HIP runtime in this process (" + g.runtime_path + ", version " + std::to_string(have) + ") is another major release than the one " +
                      core_path + " was compiled for (" + std::to_string(built) + "): rebuild the library against that ROCm, or load the matching runtime first";
            g.core = nullptr;
        }
    }
}

// More than one HIP runtime in the process (looked up when a pl_create FAILS -- the second copy may have arrived after the loader
// bound to the first): each copy keeps its own device state, and the one that initialises second finds no device.
std::string two_runtimes_note() {
    std::vector<std::string> mapped;
    dl_iterate_phdr(collect_runtimes, &mapped);
    if (mapped.size() < 2) return "";
    std::string note = "two HIP runtimes are mapped into this process: " + mapped[0];
    for (size_t i = 1; i < mapped.size(); ++i) note += " and " + mapped[i];
    return note + "; libpaule_hip is bound to " + g.runtime_path +
           " -- a device that the other copy initialised first is invisible to it (link or import ONE runtime, or import it BEFORE the first pl_* call)";
}

bool ready() {
    std::call_once(g.once, load);
    if (!g.core) t_msg = "libpaule_hip: " + g.error;
    return g.core != nullptr;
}

template <typename R>
R fail_value() { return static_cast<R>(PL_ERR_HIP); }
template <>
double fail_value<double>() { return 0.0; }
template <>
const char* fail_value<const char*>() { return t_msg.c_str(); }

}  // namespace

extern "C" {

// pl_create and pl_last_error are written out (they carry the two-runtimes note); everything else is a plain forwarder
int pl_create(const pl_config* cfg, pl_handle** out) {
    if (!ready()) return PL_ERR_HIP;
    t_msg.clear();
    const int rc = g.p_pl_create(cfg, out);
    if (rc != PL_OK) {
        const std::string note = two_runtimes_note();
        if (!note.empty()) {
            const char* m = g.p_pl_last_error();
            t_msg = std::string(m ? m : "pl_create failed") + " [" + note + "]";
        }
    }
    return rc;
}

// Neither of these two binds anything: a host may ask for the version, or for the last error of a thread that has not called anything yet,
// BEFORE it loads its own HIP runtime -- binding here would bring /opt/rocm's copy in front of it
int pl_version(void) { return PL_VERSION; }

const char* pl_last_error(void) {
    if (!g.core) return t_msg.c_str();   // nothing bound (yet, or the binding failed: ready() left its reason here)
    if (!t_msg.empty()) return t_msg.c_str();
    return g.p_pl_last_error();
}

}  // extern "C"

// the forwarders: every entry of the table except the two written out above (PL_FWD_CUSTOM in the table)
#define PL_FWD_CUSTOM(ret, name, typed, args)
#define PL_FWD(ret, name, typed, args)                    \
    extern "C" ret name typed {                           \
        if (!ready()) return fail_value<ret>();           \
        t_msg.clear();                                    \
        return g.p_##name args;                           \
    }
#include "shim_table.inc"
#undef PL_FWD
#undef PL_FWD_CUSTOM
