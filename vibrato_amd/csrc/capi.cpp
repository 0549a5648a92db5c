// extern "C" boundary of libvibrato_hip.so (see include/vibrato_hip.h).
#include <hip/hip_runtime.h>

#include <chrono>
#include <condition_variable>
#include <functional>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <algorithm>
#include <atomic>
#include <memory>
#include <mutex>
#include <stdexcept>
#include <string>
#include <thread>
#include <unordered_set>
#include <vector>

#include <pthread.h>
#include <sched.h>
#include <unistd.h>

#include <hsa/hsa.h>
#include <hsa/hsa_ext_amd.h>

#include "engine.hpp"

using namespace vbt;

struct vbt_dict {
    Dictionary* d = nullptr;
    bool owned = true;
    ~vbt_dict() { if (owned) delete d; }
};
// One pooled set of device buffers for the host-buffer entry point: a Workspace plus the text / offsets
// staging it reads, all on a private stream.  vbt_tokenize_batch takes one from its tokenizer's pool and
// puts it back, so steady-state calls allocate nothing on the device (SURVEY.md 8(b) "threading").
struct PooledWorkspace {
    std::unique_ptr<Workspace> ws;
    void* d_text = nullptr;
    uint64_t* d_off = nullptr;
    hipStream_t stream = nullptr;
    hipEvent_t done = nullptr;  // behind the last command of a chunk (pipelined host call)
    uint64_t cap_sentences = 0, cap_bytes = 0;
    ~PooledWorkspace() {
        ws.reset();
        if (done) (void)hipEventDestroy(done);
        (void)hipFree(d_text);
        (void)hipFree(d_off);
        if (stream) (void)hipStreamDestroy(stream);
    }
};

// Page-locked host memory recycled between batches: a batch's copy of the text and its results live in pinned blocks, so
// both directions are true asynchronous DMA on the workspace's stream (pageable copies are staged by the runtime and
// block), and hipHostMalloc (~ms per block) is off the steady-state path.
struct PinnedBlock {
    void* p = nullptr;
    size_t cap = 0;
    ~PinnedBlock() { if (p) (void)hipHostFree(p); }
};

// Device -> pinned-host copies on the GPU's SDMA engines (hsa_amd_memory_async_copy), next to the HIP runtime: hipMemcpyAsync
// serves a device -> pinned-host copy of this size with a copy KERNEL (and so does a kernel that stores to the mapped block
// itself); either way the 69 MB of token records of a headline batch occupy the shader memory pipes for ~1.3 ms during which
// the kernels of the other batches in flight do not advance (profiles/r03_h2h_timeline.md).  The DMA engines do not.
struct Sdma {
    hsa_agent_t gpu{}, cpu{};
    bool ok = false;
    struct Find { uint32_t bdf; uint32_t domain; hsa_agent_t gpu{}, cpu{}; bool has_gpu = false, has_cpu = false; };
    static hsa_status_t visit(hsa_agent_t a, void* data) {
        Find* f = static_cast<Find*>(data);
        hsa_device_type_t type;
        if (hsa_agent_get_info(a, HSA_AGENT_INFO_DEVICE, &type) != HSA_STATUS_SUCCESS) return HSA_STATUS_SUCCESS;
        if (type == HSA_DEVICE_TYPE_CPU) {
            if (!f->has_cpu) { f->cpu = a; f->has_cpu = true; }
        } else if (type == HSA_DEVICE_TYPE_GPU) {
            uint32_t bdf = 0, domain = 0;
            (void)hsa_agent_get_info(a, static_cast<hsa_agent_info_t>(HSA_AMD_AGENT_INFO_BDFID), &bdf);
            (void)hsa_agent_get_info(a, static_cast<hsa_agent_info_t>(HSA_AMD_AGENT_INFO_DOMAIN), &domain);
            if (!f->has_gpu && bdf == f->bdf && domain == f->domain) { f->gpu = a; f->has_gpu = true; }
        }
        return HSA_STATUS_SUCCESS;
    }
    explicit Sdma(int device) {
        hipDeviceProp_t prop;
        if (hipGetDeviceProperties(&prop, device) != hipSuccess) return;
        if (hsa_init() != HSA_STATUS_SUCCESS) return;  // (reference counted: the HIP runtime holds it already)
        Find f{(uint32_t)((prop.pciBusID << 8) | (prop.pciDeviceID << 3)), (uint32_t)prop.pciDomainID};
        if (hsa_iterate_agents(visit, &f) != HSA_STATUS_SUCCESS || !f.has_gpu || !f.has_cpu) return;
        gpu = f.gpu; cpu = f.cpu;
        // the CPU agent of the GPU's own NUMA node, where the runtime knows it (a two-socket host: the first CPU agent may be the far one)
        hsa_agent_t near{};
        if (hsa_agent_get_info(gpu, static_cast<hsa_agent_info_t>(HSA_AMD_AGENT_INFO_NEAREST_CPU), &near) == HSA_STATUS_SUCCESS && near.handle) cpu = near;
        ok = true;
        // Which SDMA engine carries the results.  Left to the runtime (hsa_amd_memory_async_copy) the choice depends on what else the
        // process has done with the device: inside an application that had used torch's copies before, the default landed on an engine
        // that moved the 68 MB of a headline batch at 29 GB/s instead of 52 -- one host call 28 M instead of 40 M sentences/s, and the
        // unexplained 18 vs 27 M box-to-box spread of round 5's text -> text leg.  So the engines are timed once, here (a 4 MiB copy,
        // best of two, on each of the first four engines the device -> host direction offers and on the default entry point), and the
        // fastest is used from then on.  VBT_SDMA_ENGINE=<bit index> picks one, -1 = the default entry point.
        uint32_t mask = 0;
        const bool have = hsa_amd_memory_copy_engine_status(cpu, gpu, &mask) == HSA_STATUS_SUCCESS;
        if (const char* e = std::getenv("VBT_SDMA_ENGINE")) {
            const int want = std::atoi(e);
            if (want >= 0 && want < 16 && have && (mask >> want & 1u)) engine = 1u << want;
        } else if (have && mask) {
            constexpr size_t kProbe = 4u << 20;
            void *d = nullptr, *hst = nullptr;
            if (hipMalloc(&d, kProbe) == hipSuccess && hipHostMalloc(&hst, kProbe, hipHostMallocPortable) == hipSuccess) {
                double best = 1e30;
                uint32_t best_engine = 0;
                for (int cand = -1; cand < 4; ++cand) {
                    if (cand >= 0 && !(mask >> cand & 1u)) continue;
                    engine = cand < 0 ? 0u : 1u << cand;
                    double t_min = 1e30;
                    for (int rep = 0; rep < 3; ++rep) {
                        std::vector<hsa_signal_t> sigs;
                        const void* src[1] = {d};
                        void* dst[1] = {hst};
                        const size_t len[1] = {kProbe};
                        const auto t0 = std::chrono::steady_clock::now();
                        const bool good = issue(src, dst, len, 1, sigs) && wait(sigs);
                        const double dt = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
                        if (!good) { t_min = 1e30; break; }
                        if (rep) t_min = std::min(t_min, dt);  // (the first copy on an engine sets its queue up)
                    }
                    if (t_min < best * 0.97) { best = t_min; best_engine = engine; }  // (the default entry point wins ties)
                }
                engine = best_engine;
                if (std::getenv("VBT_DEBUG")) std::fprintf(stderr, "[vbt] SDMA engine for results: %s%u (%.1f GB/s on the probe)\n", engine ? "bit mask " : "runtime's choice ", engine, kProbe / best / 1e9);
            }
            if (hst) (void)hipHostFree(hst);
            if (d) (void)hipFree(d);
        }
    }
    uint32_t engine = 0;  // hsa_amd_sdma_engine_id_t of the device -> host copies (0: hsa_amd_memory_async_copy picks)
    // dst[k] (pinned host) <- src[k] (device), bytes[k].  issue() starts the copies and appends one completion signal per copy
    // (initial value 1: what rocprofv3's memory-copy tracing expects of a caller, it aborts on a shared, counted signal); wait()
    // returns when all have landed.  Split in two so that the copies of several devices run side by side.
    bool issue(const void* const* src, void* const* dst, const size_t* bytes, int n, std::vector<hsa_signal_t>& sigs) const {
        for (int k = 0; k < n; ++k) {
            if (!bytes[k]) continue;
            hsa_signal_t sig;
            if (hsa_signal_create(1, 0, nullptr, &sig) != HSA_STATUS_SUCCESS) return false;
            const hsa_status_t st = engine ? hsa_amd_memory_async_copy_on_engine(dst[k], cpu, src[k], gpu, bytes[k], 0, nullptr, sig, static_cast<hsa_amd_sdma_engine_id_t>(engine), false)
                                           : hsa_amd_memory_async_copy(dst[k], cpu, src[k], gpu, bytes[k], 0, nullptr, sig);
            if (st != HSA_STATUS_SUCCESS) {
                (void)hsa_signal_destroy(sig);
                return false;
            }
            sigs.push_back(sig);
        }
        return true;
    }
    // dst (device) <- src (pinned host): the way in, next to the HIP runtime as well -- a copy the runtime enqueues on a stream waits for
    // everything in front of it in the stream's HARDWARE queue, and with the runtime's default of four hardware queues (an application
    // that initialised HIP before this library: GPU_MAX_HW_QUEUES can no longer be raised) the streams of a pipelined call's chunks
    // share queues: chunk c + 1's text then waited for chunk c's kernels and a pipelined call was slower than an unpipelined one
    bool issue_h2d(const void* src, void* dst, size_t bytes, std::vector<hsa_signal_t>& sigs) const {
        if (!bytes) return true;
        hsa_signal_t sig;
        if (hsa_signal_create(1, 0, nullptr, &sig) != HSA_STATUS_SUCCESS) return false;
        if (hsa_amd_memory_async_copy(dst, gpu, src, cpu, bytes, 0, nullptr, sig) != HSA_STATUS_SUCCESS) {
            (void)hsa_signal_destroy(sig);
            return false;
        }
        sigs.push_back(sig);
        return true;
    }
    static bool wait(std::vector<hsa_signal_t>& sigs) {
        bool good = true;
        for (hsa_signal_t sig : sigs) {
            if (!sig.handle) continue;  // (waited for and destroyed already: the pipelined call lands its chunks one by one)
            while (hsa_signal_wait_scacquire(sig, HSA_SIGNAL_CONDITION_LT, 1, UINT64_MAX, HSA_WAIT_STATE_ACTIVE) >= 1) {}
            if (hsa_signal_load_relaxed(sig) < 0) good = false;
            (void)hsa_signal_destroy(sig);
        }
        sigs.clear();
        return good;
    }
};

// One device of a tokenizer: the dictionary's device image, the idle workspaces of that device and its DMA engines.
// The CPUs of the NUMA node a device hangs off (sysfs: /sys/bus/pci/devices/<domain:bus:device.function>/numa_node, then
// /sys/devices/system/node/node<N>/cpulist); empty when the platform does not say.  The host thread that drives a device of a
// multi-device call is pinned there: it copies the shard's text into pinned memory and polls the device's signals.
static std::vector<int> cpus_near_device(int device) {
    std::vector<int> cpus;
    hipDeviceProp_t prop;
    if (hipGetDeviceProperties(&prop, device) != hipSuccess) return cpus;
    char path[128];
    std::snprintf(path, sizeof(path), "/sys/bus/pci/devices/%04x:%02x:%02x.0/numa_node", prop.pciDomainID, prop.pciBusID, prop.pciDeviceID);
    int node = -1;
    if (FILE* f = std::fopen(path, "r")) { if (std::fscanf(f, "%d", &node) != 1) node = -1; std::fclose(f); }
    if (node < 0) return cpus;
    std::snprintf(path, sizeof(path), "/sys/devices/system/node/node%d/cpulist", node);
    FILE* f = std::fopen(path, "r");
    if (!f) return cpus;
    int a = 0, b = 0;
    for (;;) {  // "0-63,128-191"
        if (std::fscanf(f, "%d", &a) != 1) break;
        b = a;
        int c = std::fgetc(f);
        if (c == '-') { if (std::fscanf(f, "%d", &b) != 1) break; c = std::fgetc(f); }
        for (int i = a; i <= b && i < CPU_SETSIZE; ++i) cpus.push_back(i);
        if (c != ',') break;
    }
    std::fclose(f);
    return cpus;
}

struct Replica {
    std::vector<int> near_cpus;                          // CPUs of the device's NUMA node (empty: unknown)
    std::unique_ptr<Tokenizer> t;
    std::vector<std::unique_ptr<PooledWorkspace>> pool;  // idle workspaces (guarded by the tokenizer's pool_mu)
    std::unique_ptr<Sdma> sdma;                          // device -> pinned-host copies on the DMA engines (created at the first host batch)
    uint64_t dev_budget = 0;                             // bytes of idle workspaces this device keeps (VBT_POOL_MAX_MB, default a quarter of its memory)
};

struct vbt_tokenizer {
    // reps[0] owns the dictionary; a multi-device tokenizer (vbt_tokenizer_new_multi) holds one replica of the device image per
    // listed device and vbt_tokenize_batch splits every batch over them.  `t` = reps[0]->t: what Worker and the device API use.
    std::vector<std::unique_ptr<Replica>> reps;
    Tokenizer* t = nullptr;
    vbt_dict dict_view;  // borrowed view handed out by vbt_tokenizer_dictionary
    std::mutex pool_mu;
    std::vector<std::unique_ptr<PinnedBlock>> host_pool;  // idle pinned blocks
    uint64_t pool_created = 0, pool_reused = 0;
    // feature strings of the three lexicons (system, user, unknown) back to back + offsets per word id: what vbt_batch_format reads
    // instead of 877 k separately allocated std::strings (built at the first format call)
    struct FlatFeatures { std::vector<uint32_t> off; std::vector<char> blob; } flat[3];
    std::once_flag flat_once;
    // tokens per KiB of text the batches so far produced at most (0: none yet): sizes the result block of a pipelined call up front
    std::atomic<uint32_t> tok_per_kib{0};
    // vbt_tokenize_batch calls running right now, calls started so far, and the last call that saw another one in flight: only a
    // caller that has been alone for a while gets its batch pipelined in chunks (host threads that stream batches side by side
    // overlap each other's copies and kernels already; a chunked call in their midst is slower for everybody)
    std::atomic<int> calls_in_flight{0};
    std::atomic<uint64_t> call_seq{0}, last_concurrent{0};
    std::atomic<int> out_mode{-1};      // VBT_H2H_OUT: 0 = the packing kernel stores into the pinned block, 1 = SDMA copies (default); published with release once the replicas' Sdma exist
    ~vbt_tokenizer() { for (auto& r : reps) r->pool.clear(); }  // before the Tokenizers (workspaces reference them)
};
struct vbt_workspace { std::unique_ptr<Workspace> w; };

struct vbt_batch {
    vbt_tokenizer* tok = nullptr;
    std::unique_ptr<PinnedBlock> in_blk, out_blk;  // go back to the tokenizer's pool in vbt_batch_free
    uint64_t n = 0, n_tokens = 0;
    const uint64_t* offsets = nullptr;  // in_blk: n + 1 rebased offsets, then the text
    const uint8_t* text = nullptr;
    const uint32_t* tok_off = nullptr;  // out_blk: tok_off[n], tok_cnt[n], tokens[n_tokens]
    const uint32_t* tok_cnt = nullptr;
    const vbt_token_rec* tokens = nullptr;
};

struct vbt_worker {
    const vbt_tokenizer* tok;
    std::string text;
    std::unique_ptr<Workspace> ws;
    void* d_text = nullptr;
    uint64_t* d_offsets = nullptr;
    size_t cap = 0;  // bytes the workspace, d_text and the pinned block below hold
    // latency path (Workspace::serve): one pinned host block {text, padded to 16 | 16 control words | token records}, read and written
    // by a resident kernel directly (tokenize_serve in engine.hip: control word layout), and the worker's own stream, which that
    // kernel occupies while it is resident
    void* h_block = nullptr;
    uint8_t* h_text = nullptr;
    uint32_t* h_ctl = nullptr;  // kernel: [0] sequence served, [1] token count, [2] has left, [3] outcome; host: [4] doorbell, [5] bytes, [6] leave now
    vbt_token_rec* h_tokens = nullptr;
    uint8_t* hd_text = nullptr;  // device addresses of the same
    uint32_t* hd_ctl = nullptr;
    vbt_token_rec* hd_tokens = nullptr;
    hipStream_t stream = nullptr;
    int single = -1;   // VBT_WORKER_SINGLE=0: every sentence through the batch pipeline (round-2 behaviour, kept for A/B)
    uint32_t idle_polls = 2000;  // VBT_WORKER_IDLE_POLLS: polls without a call after which the resident kernel leaves (~2 ms); 0 = one launch per call (round 3)
    uint32_t seq = 0;       // sequence number of the last call handed to the kernel
    bool serving = false;   // a resident kernel was started and has not been seen to leave
    uint64_t n_fast = 0, n_slow = 0, n_launches = 0;  // sentences served by the resident kernel / handed to the batch pipeline; kernels started
    // Tells a resident kernel to leave and waits for it (before the stream or the buffers it uses are touched by anything else).
    void stop_serving() {
        if (!serving) return;
        __atomic_store_n(&h_ctl[6], 1u, __ATOMIC_RELEASE);
        (void)hipStreamSynchronize(stream);
        h_ctl[6] = 0; h_ctl[2] = 0;
        serving = false;
    }
    std::vector<vbt_token_rec> tokens;
    // ConnIdCounter of Worker::init_connid_counter (worker.rs:77-84, mapper.rs:87-106); empty = never initialised
    std::vector<uint64_t> lid_count, rid_count;
    void release() {
        stop_serving();
        ws.reset();
        (void)hipFree(d_text); (void)hipFree(d_offsets);
        if (h_block) (void)hipHostFree(h_block);
        d_text = nullptr; d_offsets = nullptr; h_block = nullptr;
    }
    ~vbt_worker() {
        release();
        if (stream) (void)hipStreamDestroy(stream);
    }
};

namespace {

thread_local std::string g_last_error;

template <typename F>
int guarded(F&& f) {
    try {
        f();
        return VBT_OK;
    } catch (const Error& e) {
        g_last_error = e.what();
        return e.code;
    } catch (const std::bad_alloc&) {
        g_last_error = "out of host memory";
        return VBT_ERR_INVALID_STATE;
    } catch (const std::exception& e) {
        g_last_error = e.what();
        return VBT_ERR_INVALID_STATE;
    }
}

#define HIPX(expr)                                                                                  \
    do {                                                                                            \
        hipError_t e_ = (expr);                                                                     \
        if (e_ != hipSuccess) throw Error(VBT_ERR_DEVICE, std::string(#expr) + ": " + hipGetErrorString(e_)); \
    } while (0)

const Lexicon& lexicon_of(const Dictionary& d, uint32_t lex_type) {
    if (lex_type == VBT_LEX_SYSTEM) return d.system;
    if (lex_type == VBT_LEX_USER && d.has_user) return d.user;
    throw Error(VBT_ERR_INVALID_ARGUMENT, "lex_type: no such lexicon");
}

void check_device_errors(uint32_t flags) {
    if (flags & kErrOffsets) throw Error(VBT_ERR_INVALID_ARGUMENT, "offsets must be non-decreasing and span at most total_bytes");
    if (flags & kErrUtf8) throw Error(VBT_ERR_UTF8, "sentence text is not valid UTF-8");
    if (flags & kErrTokCap) throw Error(VBT_ERR_INVALID_STATE, "device token buffer overflow");
    if (flags & kErrScratch) throw Error(VBT_ERR_UNSUPPORTED, "device scratch arena exhausted (raise VBT_SCRATCH_MB)");
    if (flags & kErrTooLong) throw Error(VBT_ERR_UNSUPPORTED, "sentence too long for the device image (>= 4 GiB or >= 2^32 lattice nodes)");
}

// Token accessors (token.rs:21-92) from a raw record.
void fill_token(const Dictionary& d, const uint8_t* sentence, const vbt_token_rec& r, vbt_token* out) {
    const uint32_t lex = r.word_idx >> 30, wid = r.word_idx & 0x3FFFFFFFu;
    out->surface = reinterpret_cast<const char*>(sentence) + r.start_byte;
    out->surface_len = r.end_byte - r.start_byte;
    out->start_char = r.start_char; out->end_char = r.end_char;
    out->start_byte = r.start_byte; out->end_byte = r.end_byte;
    out->lex_type = lex; out->word_id = wid;
    out->total_cost = r.total_cost;
    if (lex == VBT_LEX_UNKNOWN) {
        const Entry& e = d.unk_entries.at(wid);
        out->left_id = (uint16_t)(e.left_right & 0xFFFF); out->right_id = (uint16_t)(e.left_right >> 16);
        out->word_cost = (int16_t)(uint16_t)e.cost;
        out->feature = d.unk_features[wid].data(); out->feature_len = d.unk_features[wid].size();
    } else {
        const Lexicon& lx = lex == VBT_LEX_SYSTEM ? d.system : d.user;
        const WordParam& p = lx.params.at(wid);
        out->left_id = p.left_id; out->right_id = p.right_id; out->word_cost = p.word_cost;
        out->feature = lx.features[wid].data(); out->feature_len = lx.features[wid].size();
    }
}

// Runs one batch through a workspace and copies the compact results to the host.
void run_and_fetch(Workspace& ws, const uint8_t* d_text, const uint64_t* d_off, uint64_t n, uint64_t bytes,
                   std::vector<uint32_t>& tok_off, std::vector<uint32_t>& tok_cnt, std::vector<vbt_token_rec>& tokens,
                   hipStream_t stream = nullptr) {
    ws.run(d_text, d_off, n, bytes, stream);
    vbt_call_stats st;
    ws.stats(&st);
    check_device_errors(st.error_flags);
    tok_off.resize(n); tok_cnt.resize(n); tokens.resize(st.n_tokens);
    if (n) {
        HIPX(hipMemcpy(tok_off.data(), ws.d_tok_off, n * 4, hipMemcpyDeviceToHost));
        HIPX(hipMemcpy(tok_cnt.data(), ws.d_tok_cnt, n * 4, hipMemcpyDeviceToHost));
    }
    if (st.n_tokens) HIPX(hipMemcpy(tokens.data(), ws.d_tokens, st.n_tokens * sizeof(vbt_token_rec), hipMemcpyDeviceToHost));
}

uint64_t round_up_pow2(uint64_t v, uint64_t lo) {
    uint64_t r = lo;
    while (r < v) r <<= 1;
    return r;
}

// Smallest idle workspace of the replica that holds the request, else a new one (capacities rounded up to powers of two so
// that batches of similar size share it).  At most kPoolIdle workspaces stay idle per device; the smallest is dropped first.
constexpr size_t kPoolIdle = 16;
std::unique_ptr<PooledWorkspace> pool_take(vbt_tokenizer* tok, Replica& rep, uint64_t n, uint64_t bytes) {
    {
        std::lock_guard<std::mutex> g(tok->pool_mu);
        size_t best = rep.pool.size();
        for (size_t i = 0; i < rep.pool.size(); ++i) {
            const auto& p = rep.pool[i];
            if (p->cap_sentences >= n && p->cap_bytes >= bytes && (best == rep.pool.size() || p->cap_bytes < rep.pool[best]->cap_bytes)) best = i;
        }
        if (best != rep.pool.size()) {
            auto p = std::move(rep.pool[best]);
            rep.pool.erase(rep.pool.begin() + (long)best);
            ++tok->pool_reused;
            return p;
        }
        ++tok->pool_created;
    }
    HIPX(hipSetDevice(rep.t->device()));
    auto p = std::make_unique<PooledWorkspace>();
    p->cap_sentences = round_up_pow2(n, 64);
    p->cap_bytes = round_up_pow2(bytes, 4096);
    p->ws = std::make_unique<Workspace>(*rep.t, p->cap_sentences, p->cap_bytes);
    HIPX(hipMalloc(&p->d_text, p->cap_bytes));
    HIPX(hipMalloc(reinterpret_cast<void**>(&p->d_off), (p->cap_sentences + 1) * 8));
    HIPX(hipStreamCreateWithFlags(&p->stream, hipStreamNonBlocking));
    HIPX(hipEventCreateWithFlags(&p->done, hipEventDisableTiming));
    return p;
}

constexpr size_t kHostPoolIdle = 32;  // (two per batch in flight: 8 made six host threads free and re-pin their blocks all the time)
std::unique_ptr<PinnedBlock> host_take(vbt_tokenizer* tok, size_t bytes) {
    {
        std::lock_guard<std::mutex> g(tok->pool_mu);
        size_t best = tok->host_pool.size();
        for (size_t i = 0; i < tok->host_pool.size(); ++i)
            if (tok->host_pool[i]->cap >= bytes && (best == tok->host_pool.size() || tok->host_pool[i]->cap < tok->host_pool[best]->cap)) best = i;
        if (best != tok->host_pool.size()) {
            auto b = std::move(tok->host_pool[best]);
            tok->host_pool.erase(tok->host_pool.begin() + (long)best);
            return b;
        }
    }
    auto b = std::make_unique<PinnedBlock>();
    b->cap = round_up_pow2(bytes, 1 << 16);
    HIPX(hipHostMalloc(&b->p, b->cap, hipHostMallocPortable | hipHostMallocMapped));  // (reachable from every device of a multi-device tokenizer)
    return b;
}

// Budgets of the idle pools (capacities are rounded up to powers of two and were never trimmed: a few large calls
// pinned tens of GiB for the tokenizer's lifetime).  VBT_POOL_MAX_MB=<device MB>[,<pinned MB>]; the device budget is per device.
// device default: a quarter of that GPU's memory (72 GiB on an MI355X: eight host threads streaming headline-sized batches
// hold 58 GiB of workspaces, and a workspace that does not fit the budget is freed and re-allocated on every call -- with
// the fixed 32 GiB of before, six threads ran at 0.6x and eight at 0.07x the throughput of four)
uint64_t device_pool_budget() {  // of the CURRENT device (called once per replica, right after its image was uploaded)
    unsigned long long d = 32768;
    size_t free_b = 0, total_b = 0;
    if (hipMemGetInfo(&free_b, &total_b) == hipSuccess && (total_b >> 22) > d) d = total_b >> 22;
    if (const char* e = std::getenv("VBT_POOL_MAX_MB")) {
        unsigned long long a = 0, b = 0;
        if (std::sscanf(e, "%llu,%llu", &a, &b) >= 1) d = a;
    }
    return (uint64_t)d << 20;
}
uint64_t host_pool_budget() {
    unsigned long long h = 8192;
    if (const char* e = std::getenv("VBT_POOL_MAX_MB")) {
        unsigned long long a = 0, b = 0;
        if (std::sscanf(e, "%llu,%llu", &a, &b) >= 2) h = b;
    }
    return (uint64_t)h << 20;
}
uint64_t workspace_bytes(const PooledWorkspace& p) { return 400 * p.cap_bytes + 7200 * p.cap_sentences; }  // (vibrato_hip.h: footprint)

void host_give(vbt_tokenizer* tok, std::unique_ptr<PinnedBlock> b) {
    if (!b) return;
    static const uint64_t host_budget = host_pool_budget();
    std::unique_ptr<PinnedBlock> drop;  // hipHostFree synchronises the device: outside the lock
    {
        std::lock_guard<std::mutex> g(tok->pool_mu);
        uint64_t held = b->cap;
        for (const auto& q : tok->host_pool) held += q->cap;
        if (held > host_budget) drop = std::move(b);
        else {
            tok->host_pool.push_back(std::move(b));
            if (tok->host_pool.size() > kHostPoolIdle) {
                size_t smallest = 0;
                for (size_t i = 1; i < tok->host_pool.size(); ++i)
                    if (tok->host_pool[i]->cap < tok->host_pool[smallest]->cap) smallest = i;
                drop = std::move(tok->host_pool[smallest]);
                tok->host_pool.erase(tok->host_pool.begin() + (long)smallest);
            }
        }
    }
}

void pool_give(vbt_tokenizer* tok, Replica& rep, std::unique_ptr<PooledWorkspace> p) {
    std::unique_ptr<PooledWorkspace> drop;  // destroyed outside the lock
    {
        std::lock_guard<std::mutex> g(tok->pool_mu);
        uint64_t held = workspace_bytes(*p);
        for (const auto& q : rep.pool) held += workspace_bytes(*q);
        if (held > rep.dev_budget) drop = std::move(p);
        else {
            rep.pool.push_back(std::move(p));
            if (rep.pool.size() > kPoolIdle) {
                size_t smallest = 0;
                for (size_t i = 1; i < rep.pool.size(); ++i)
                    if (rep.pool[i]->cap_bytes < rep.pool[smallest]->cap_bytes) smallest = i;
                drop = std::move(rep.pool[smallest]);
                rep.pool.erase(rep.pool.begin() + (long)smallest);
            }
        }
    }
    if (drop) { (void)hipSetDevice(rep.t->device()); drop.reset(); }
}

// Buffers the library hands out and vbt_free takes back (vbt_batch_format's text, vbt_dict_write's bytes).  A 16-byte header in
// front of the caller's pointer holds the capacity; the last few big ones are kept instead of freed: 100 MB of `tokenize` output per
// batch would otherwise be mapped, faulted in page by page (the kernel zeroes every page first: that, not the copying, bounded the
// formatter at ~10 GB/s on 64 threads) and unmapped again for every batch.
// Only pointers this library handed out are ever touched: the live ones are registered, so a second vbt_free of the same pointer or
// a pointer from somewhere else is IGNORED -- nothing in front of a foreign pointer is read, no block can enter the cache twice
// (round-5 advisor).  The cache holds at most kOutCacheSlots buffers / kOutCacheMaxTotal bytes and is released by
// vbt_tokenizer_trim_pool and vbt_tokenizer_free.
struct OutHeader { uint64_t cap, pad; };
constexpr size_t kOutCacheSlots = 4;
constexpr uint64_t kOutCacheMinBytes = 1u << 20, kOutCacheMaxTotal = 512ull << 20;
std::mutex g_out_mu;
std::vector<OutHeader*> g_out_cache;        // idle buffers
std::unordered_set<const void*> g_out_live;  // the caller-side pointers of the buffers that are handed out

void* out_alloc(size_t bytes) {
    OutHeader* h = nullptr;
    if (bytes >= kOutCacheMinBytes) {
        std::lock_guard<std::mutex> g(g_out_mu);
        size_t best = g_out_cache.size();
        for (size_t i = 0; i < g_out_cache.size(); ++i)
            if (g_out_cache[i]->cap >= bytes && g_out_cache[i]->cap <= 2 * bytes + (1u << 20) && (best == g_out_cache.size() || g_out_cache[i]->cap < g_out_cache[best]->cap)) best = i;
        if (best != g_out_cache.size()) {
            h = g_out_cache[best];
            g_out_cache.erase(g_out_cache.begin() + (long)best);
            g_out_live.insert(h + 1);
            return h + 1;
        }
    }
    const size_t cap = bytes >= kOutCacheMinBytes ? bytes + bytes / 8 : (bytes ? bytes : 1);  // (head room: the next batch's output is about as long)
    h = static_cast<OutHeader*>(std::malloc(sizeof(OutHeader) + cap));
    if (!h) return nullptr;
    h->cap = cap; h->pad = 0;
    std::lock_guard<std::mutex> g(g_out_mu);
    g_out_live.insert(h + 1);
    return h + 1;
}

void out_free(void* p) {
    if (!p) return;
    OutHeader* h = nullptr;
    {
        std::lock_guard<std::mutex> g(g_out_mu);
        if (g_out_live.erase(p) == 0) return;  // not a live buffer of this library: freed already, or never ours
        h = static_cast<OutHeader*>(p) - 1;
        if (h->cap >= kOutCacheMinBytes) {
            uint64_t held = h->cap;
            for (const OutHeader* q : g_out_cache) held += q->cap;
            if (g_out_cache.size() < kOutCacheSlots && held <= kOutCacheMaxTotal) { g_out_cache.push_back(h); return; }
        }
    }
    std::free(h);
}

void out_cache_trim() {
    std::vector<OutHeader*> drop;
    {
        std::lock_guard<std::mutex> g(g_out_mu);
        drop.swap(g_out_cache);
    }
    for (OutHeader* h : drop) std::free(h);
}

// Persistent host threads for the formatter (starting 63 threads per call cost 2-3 ms of a 10 ms call).  run(T, body): body(k) for
// k = 0 .. T - 1, k = 0 on the calling thread, the others on pool threads (created on demand, parked on a condition variable
// between calls); returns how many chunks ran concurrently = T, or fewer if the host refused more threads (then the caller is told
// BEFORE any body runs, through `granted`).  One job at a time (callers serialise on job_mu): the formatter is memory-bound, two
// at once gain nothing.
class HostPool {
  public:
    // (leaked on purpose: parked threads at process exit.  One pool per PROCESS: a fork()ed child inherits the object with its thread
    // count but none of the threads -- it would wait for workers that do not exist -- so the child gets a pool of its own.)
    static HostPool& get() {
        static std::mutex mu;
        static HostPool* p = nullptr;
        static pid_t owner = 0;
        std::lock_guard<std::mutex> g(mu);
        if (!p || owner != getpid()) { p = new HostPool; owner = getpid(); }
        return *p;
    }
    // reserves up to `want` workers (this thread included); returns the number granted; run() must follow with exactly that T
    unsigned begin(unsigned want) {
        job_mu_.lock();
        std::lock_guard<std::mutex> g(mu_);
        while (threads_ + 1 < want) {
            try { std::thread([this, idx = threads_ + 1] { loop(idx); }).detach(); ++threads_; }
            catch (const std::exception&) { break; }
        }
        return std::min<unsigned>(want, threads_ + 1);
    }
    // (`fn` is built by the caller BEFORE begin(): nothing between begin() and the hand-over below can throw with job_mu_ held)
    void run(unsigned T, std::function<void(unsigned)>& fn) {
        {
            std::lock_guard<std::mutex> g(mu_);
            fn_ = &fn; active_ = T; pending_ = T - 1; ++gen_;
        }
        cv_.notify_all();
        fn(0);
        std::unique_lock<std::mutex> g(mu_);
        done_cv_.wait(g, [&] { return pending_ == 0; });
        fn_ = nullptr;
        g.unlock();
        job_mu_.unlock();
    }
    void cancel() { job_mu_.unlock(); }  // begin() without run()

  private:
    void loop(unsigned idx) {
        uint64_t seen = 0;
        for (;;) {
            std::function<void(unsigned)>* fn;
            {
                std::unique_lock<std::mutex> g(mu_);
                cv_.wait(g, [&] { return gen_ != seen; });
                seen = gen_;
                if (idx >= active_) continue;
                fn = fn_;
            }
            (*fn)(idx);
            std::lock_guard<std::mutex> g(mu_);
            if (--pending_ == 0) done_cv_.notify_all();
        }
    }
    std::mutex mu_, job_mu_;
    std::condition_variable cv_, done_cv_;
    std::function<void(unsigned)>* fn_ = nullptr;
    unsigned threads_ = 0, active_ = 0, pending_ = 0;
    uint64_t gen_ = 0;
};

const Dictionary& dict_of(const vbt_dict* dict) {
    if (!dict || !dict->d) throw Error(VBT_ERR_INVALID_ARGUMENT, "dict: null or consumed by vbt_tokenizer_new");
    return *dict->d;
}

}  // namespace

extern "C" {

const char* vbt_last_error(void) { return g_last_error.c_str(); }

int vbt_utf8_valid(const char* utf8, size_t len) { return (!len || (utf8 && valid_utf8(reinterpret_cast<const uint8_t*>(utf8), len))) ? 1 : 0; }

int vbt_dict_from_sources(const char* lex, size_t lex_len, const char* matrix_def, size_t matrix_len, const char* char_def,
                          size_t char_len, const char* unk_def, size_t unk_len, vbt_dict** out) {
    return guarded([&] {
        if (!out || !lex || !matrix_def || !char_def || !unk_def) throw Error(VBT_ERR_INVALID_ARGUMENT, "null argument");
        Dictionary* d = build_dictionary({lex, lex_len}, {matrix_def, matrix_len}, nullptr, 0, 0, {char_def, char_len}, {unk_def, unk_len});
        *out = new vbt_dict{d, true};
    });
}

int vbt_dict_from_sources_binmatrix(const char* lex, size_t lex_len, const int16_t* matrix, uint32_t num_right, uint32_t num_left,
                                    const char* char_def, size_t char_len, const char* unk_def, size_t unk_len, vbt_dict** out) {
    return guarded([&] {
        if (!out || !lex || !matrix || !char_def || !unk_def) throw Error(VBT_ERR_INVALID_ARGUMENT, "null argument");
        Dictionary* d = build_dictionary({lex, lex_len}, {}, matrix, num_right, num_left, {char_def, char_len}, {unk_def, unk_len});
        *out = new vbt_dict{d, true};
    });
}

int vbt_dict_from_sources_bigram(const char* lex, size_t lex_len, const char* bigram_right, size_t right_len, const char* bigram_left,
                                 size_t left_len, const char* bigram_cost, size_t cost_len, const char* char_def, size_t char_len,
                                 const char* unk_def, size_t unk_len, int dual, vbt_dict** out) {
    return guarded([&] {
        if (!out || !lex || !bigram_right || !bigram_left || !bigram_cost || !char_def || !unk_def) throw Error(VBT_ERR_INVALID_ARGUMENT, "null argument");
        Dictionary* d = build_dictionary_bigram({lex, lex_len}, {bigram_right, right_len}, {bigram_left, left_len}, {bigram_cost, cost_len},
                                                {char_def, char_len}, {unk_def, unk_len}, dual != 0);
        *out = new vbt_dict{d, true};
    });
}

int vbt_dict_read(const uint8_t* data, size_t len, vbt_dict** out) {
    return guarded([&] {
        if (!out || (!data && len)) throw Error(VBT_ERR_INVALID_ARGUMENT, "null argument");
        *out = new vbt_dict{read_dictionary(data, len), true};
    });
}

int vbt_dict_write(const vbt_dict* dict, int zstd_level, uint8_t** out, size_t* len) {
    return guarded([&] {
        if (!out || !len) throw Error(VBT_ERR_INVALID_ARGUMENT, "null argument");
        std::vector<uint8_t> bytes = write_dictionary(dict_of(dict));
        if (zstd_level >= 0) bytes = zstd_compress(bytes.data(), bytes.size(), zstd_level);
        uint8_t* p = static_cast<uint8_t*>(out_alloc(bytes.size()));
        if (!p) throw std::bad_alloc();
        std::memcpy(p, bytes.data(), bytes.size());
        *out = p;
        *len = bytes.size();
    });
}

int vbt_dict_connector_kind(const vbt_dict* dict) { return dict && dict->d ? dict->d->conn_kind : -1; }

int vbt_dict_set_user_lexicon(vbt_dict* dict, const char* csv, size_t len) {
    return guarded([&] {
        if (!dict || !dict->d) throw Error(VBT_ERR_INVALID_ARGUMENT, "dict: null or consumed");
        set_user_lexicon(*dict->d, csv, len);
    });
}

int vbt_dict_map_connection_ids(vbt_dict* dict, const uint16_t* lmap, size_t n_lmap, const uint16_t* rmap, size_t n_rmap) {
    return guarded([&] {
        if (!dict || !dict->d || !dict->owned) throw Error(VBT_ERR_INVALID_ARGUMENT, "dict: null or consumed");
        if ((!lmap && n_lmap) || (!rmap && n_rmap)) throw Error(VBT_ERR_INVALID_ARGUMENT, "null argument");
        map_connection_ids(*dict->d, lmap, n_lmap, rmap, n_rmap);
    });
}

void vbt_dict_free(vbt_dict* dict) { delete dict; }

uint32_t vbt_dict_num_words(const vbt_dict* dict, uint32_t lex_type) {
    if (!dict || !dict->d) return 0;
    const Dictionary& d = *dict->d;
    if (lex_type == VBT_LEX_SYSTEM) return (uint32_t)d.system.params.size();
    if (lex_type == VBT_LEX_USER) return d.has_user ? (uint32_t)d.user.params.size() : 0;
    return (uint32_t)d.unk_entries.size();
}
uint32_t vbt_dict_num_left(const vbt_dict* dict) { return dict && dict->d ? dict->d->num_left : 0; }
uint32_t vbt_dict_num_right(const vbt_dict* dict) { return dict && dict->d ? dict->d->num_right : 0; }

int vbt_dict_word_feature(const vbt_dict* dict, uint32_t lex_type, uint32_t word_id, const char** ptr, size_t* len) {
    return guarded([&] {
        const Dictionary& d = dict_of(dict);
        const std::string* s;
        if (lex_type == VBT_LEX_UNKNOWN) {
            if (word_id >= d.unk_features.size()) throw Error(VBT_ERR_INVALID_ARGUMENT, "word_id out of range");
            s = &d.unk_features[word_id];
        } else {
            const Lexicon& lx = lexicon_of(d, lex_type);
            if (word_id >= lx.features.size()) throw Error(VBT_ERR_INVALID_ARGUMENT, "word_id out of range");
            s = &lx.features[word_id];
        }
        *ptr = s->data();
        *len = s->size();
    });
}

int vbt_dict_word_param(const vbt_dict* dict, uint32_t lex_type, uint32_t word_id, int32_t out[3]) {
    return guarded([&] {
        const Dictionary& d = dict_of(dict);
        if (lex_type == VBT_LEX_UNKNOWN) {
            if (word_id >= d.unk_entries.size()) throw Error(VBT_ERR_INVALID_ARGUMENT, "word_id out of range");
            const Entry& e = d.unk_entries[word_id];
            out[0] = e.left_right & 0xFFFF; out[1] = e.left_right >> 16; out[2] = (int16_t)(uint16_t)e.cost;
        } else {
            const Lexicon& lx = lexicon_of(d, lex_type);
            if (word_id >= lx.params.size()) throw Error(VBT_ERR_INVALID_ARGUMENT, "word_id out of range");
            out[0] = lx.params[word_id].left_id; out[1] = lx.params[word_id].right_id; out[2] = lx.params[word_id].word_cost;
        }
    });
}

int vbt_dict_conn_cost(const vbt_dict* dict, uint32_t right_id, uint32_t left_id, int32_t* out) {
    return guarded([&] {
        const Dictionary& d = dict_of(dict);
        if (right_id >= d.num_right || left_id >= d.num_left) throw Error(VBT_ERR_INVALID_ARGUMENT, "connection id out of range");
        *out = conn_cost(d, right_id, left_id);
    });
}

uint32_t vbt_dict_char_info(const vbt_dict* dict, uint32_t cp) { return dict && dict->d ? dict->d->chr2inf[cp < 65536 ? cp : 0] : 0; }

int vbt_dict_cate_id(const vbt_dict* dict, const char* name, size_t len) { return dict && dict->d && name ? dict->d->cate_id({name, len}) : -1; }

uint32_t vbt_dict_common_prefix(const vbt_dict* dict, uint32_t lex_type, const uint32_t* cps, uint32_t n, uint32_t* out, uint32_t cap) {
    if (!dict || !dict->d) return 0;
    const Dictionary& d = *dict->d;
    if (lex_type == VBT_LEX_USER && !d.has_user) return 0;
    const Lexicon& lx = lex_type == VBT_LEX_USER ? d.user : d.system;
    std::vector<std::pair<uint32_t, uint32_t>> m;
    lx.common_prefix(cps, n, m);
    for (uint32_t i = 0; i < m.size() && i < cap; ++i) { out[2 * i] = m[i].first; out[2 * i + 1] = m[i].second; }
    return (uint32_t)m.size();
}

int vbt_tokenizer_new_multi(vbt_dict* dict, int ignore_space, uint32_t max_grouping_len, const int* devices, uint32_t n_devices, vbt_tokenizer** out) {
    return guarded([&] {
        if (!dict || !dict->d || !dict->owned || !out) throw Error(VBT_ERR_INVALID_ARGUMENT, "dict: null or already consumed");
        if (!devices || n_devices == 0 || n_devices > 64) throw Error(VBT_ERR_INVALID_ARGUMENT, "devices: expected 1..64 HIP device indices");
        // Tokenizer::new moves the dictionary in (tokenizer.rs:26). On failure the caller keeps the handle.
        auto h = std::make_unique<vbt_tokenizer>();
        for (uint32_t k = 0; k < n_devices; ++k) {
            auto rep = std::make_unique<Replica>();
            rep->t = std::make_unique<Tokenizer>(dict->d, ignore_space != 0, max_grouping_len, devices[k]);  // (leaves that device current)
            rep->dev_budget = device_pool_budget();
            rep->near_cpus = cpus_near_device(rep->t->device());
            h->reps.push_back(std::move(rep));
        }
        h->reps[0]->t->adopt(std::unique_ptr<Dictionary>(dict->d));  // the other replicas borrow it: reps[0] is destroyed last
        dict->d = nullptr;
        h->t = h->reps[0]->t.get();
        h->dict_view.d = const_cast<Dictionary*>(&h->t->dict());
        h->dict_view.owned = false;
        *out = h.release();
        delete dict;
    });
}

int vbt_tokenizer_new(vbt_dict* dict, int ignore_space, uint32_t max_grouping_len, int device, vbt_tokenizer** out) {
    return vbt_tokenizer_new_multi(dict, ignore_space, max_grouping_len, &device, 1, out);
}

uint32_t vbt_tokenizer_num_devices(const vbt_tokenizer* tok) { return tok ? (uint32_t)tok->reps.size() : 0; }

int vbt_tokenizer_connid_reorder_info(const vbt_tokenizer* tok, uint64_t out[8]) {
    return guarded([&] {
        if (!tok || !out) throw Error(VBT_ERR_INVALID_ARGUMENT, "null argument");
        const ConnidReorderInfo r = tok->t->reorder_info();
        out[0] = r.epoch; out[1] = r.state; out[2] = r.sample_sentences; out[3] = r.min_sentences;
        out[4] = (uint64_t)(r.ms * 1e3); out[5] = r.moved_left; out[6] = r.moved_right; out[7] = 0;
    });
}

int vbt_tokenizer_calibrate(const vbt_tokenizer* tok, const uint8_t* text, const uint64_t* offsets, uint64_t n) {
    return guarded([&] {
        if (!tok || !offsets || (!text && n && offsets[n] != offsets[0])) throw Error(VBT_ERR_INVALID_ARGUMENT, "null argument");
        for (auto& r : tok->reps) r->t->calibrate_host(text, offsets, n);
    });
}

int vbt_tokenizer_connid_reorder_wait(const vbt_tokenizer* tok, int64_t timeout_ms, int* idle) {
    return guarded([&] {
        if (!tok) throw Error(VBT_ERR_INVALID_ARGUMENT, "null argument");
        bool all = true;
        for (auto& r : tok->reps) all = r->t->wait_calibration(timeout_ms) && all;
        if (idle) *idle = all ? 1 : 0;
    });
}

int vbt_tokenizer_lattice_density(const vbt_tokenizer* tok, double* candidates_per_byte) {
    return guarded([&] {
        if (!tok || !candidates_per_byte) throw Error(VBT_ERR_INVALID_ARGUMENT, "null argument");
        *candidates_per_byte = tok->reps[0]->t->candidates_per_byte();
    });
}

void vbt_tokenizer_free(vbt_tokenizer* tok) {
    if (!tok) return;
    for (auto& r : tok->reps) r->pool.clear();
    while (tok->reps.size() > 1) tok->reps.pop_back();  // the replicas that borrow the dictionary go first
    delete tok;
    out_cache_trim();
}

const vbt_dict* vbt_tokenizer_dictionary(const vbt_tokenizer* tok) { return &tok->dict_view; }  // borrowed: do not free

int vbt_worker_new(const vbt_tokenizer* tok, vbt_worker** out) {
    return guarded([&] {
        if (!tok || !out) throw Error(VBT_ERR_INVALID_ARGUMENT, "null argument");
        auto w = std::make_unique<vbt_worker>();
        w->tok = tok;
        *out = w.release();
    });
}

void vbt_worker_free(vbt_worker* w) { delete w; }

int vbt_worker_reset_sentence(vbt_worker* w, const char* utf8, size_t len) {
    return guarded([&] {
        if (!w) throw Error(VBT_ERR_INVALID_ARGUMENT, "null argument");
        w->tokens.clear();
        // the reference takes &str (worker.rs:34): anything else is rejected here instead of being decoded as garbage
        if (utf8 && len && !valid_utf8(reinterpret_cast<const uint8_t*>(utf8), len)) {
            w->text.clear();
            throw Error(VBT_ERR_UTF8, "sentence is not valid UTF-8");
        }
        w->text.assign(utf8 ? utf8 : "", utf8 ? len : 0);
    });
}

// Worker::tokenize (worker.rs:49-55).  No launch per sentence: a resident kernel (Workspace::serve / tokenize_serve) reads the text
// from the worker's pinned block when the doorbell rings and writes the token records back into it; steady state allocates nothing,
// launches nothing and issues no copy.  The kernel leaves after ~2 ms without a call (it must not sit on the GPU of an idle
// process: anything that synchronises the device would wait for it) and is started again by the next call.  Sentences the single
// wavefront cannot take (outcome 1) and workers that count connection ids go through the batch pipeline.
int vbt_worker_tokenize(vbt_worker* w) {
    return guarded([&] {
        if (!w) throw Error(VBT_ERR_INVALID_ARGUMENT, "null argument");
        w->tokens.clear();
        if (w->text.empty()) return;  // worker.rs:50-52
        const size_t len = w->text.size();
        if (len >= 0xFFFFFFF0ull) throw Error(VBT_ERR_INVALID_ARGUMENT, "sentence too large");
        HIPX(hipSetDevice(w->tok->t->device()));
        if (!w->ws || w->cap < len) {
            w->release();  // (stops a resident kernel first)
            const size_t cap = (std::max<size_t>(len * 2, 4096) + 15) & ~(size_t)15;
            w->ws = std::make_unique<Workspace>(*w->tok->t, 1, cap);
            if (!w->lid_count.empty()) w->ws->enable_connid_counts(true);
            HIPX(hipMalloc(&w->d_text, cap + 16));
            HIPX(hipMalloc(reinterpret_cast<void**>(&w->d_offsets), 16));
            const size_t ctl_off = cap, tok_off = cap + 64;
            HIPX(hipHostMalloc(&w->h_block, tok_off + (cap + 2) * sizeof(vbt_token_rec), hipHostMallocDefault));
            void* dev = nullptr;
            HIPX(hipHostGetDevicePointer(&dev, w->h_block, 0));
            w->h_text = static_cast<uint8_t*>(w->h_block);
            w->h_ctl = reinterpret_cast<uint32_t*>(w->h_text + ctl_off);
            w->h_tokens = reinterpret_cast<vbt_token_rec*>(w->h_text + tok_off);
            w->hd_text = static_cast<uint8_t*>(dev);
            w->hd_ctl = reinterpret_cast<uint32_t*>(w->hd_text + ctl_off);
            w->hd_tokens = reinterpret_cast<vbt_token_rec*>(w->hd_text + tok_off);
            std::memset(w->h_ctl, 0, 64);
            w->h_ctl[0] = w->h_ctl[4] = w->seq;
            if (!w->stream) HIPX(hipStreamCreateWithFlags(&w->stream, hipStreamNonBlocking));
            if (w->single < 0) {
                const char* e = std::getenv("VBT_WORKER_SINGLE");
                w->single = e && *e ? std::atoi(e) : 1;
                e = std::getenv("VBT_WORKER_IDLE_POLLS");
                if (e && *e) w->idle_polls = (uint32_t)std::strtoul(e, nullptr, 10);
            }
            w->cap = cap;
        }
        const bool counting = !w->lid_count.empty();
        if (!counting && !w->ws->fused && w->single) {
            std::memcpy(w->h_text, w->text.data(), len);
            volatile uint32_t* ctl = w->h_ctl;
            const uint32_t seq = ++w->seq;
            ctl[5] = (uint32_t)len;
            __atomic_store_n(&w->h_ctl[4], seq, __ATOMIC_RELEASE);  // the doorbell: behind the text and its length
            auto start = [&] {  // a kernel that serves from sequence number seq on
                w->h_ctl[2] = 0;
                w->ws->serve(w->hd_text, static_cast<uint8_t*>(w->d_text), w->d_offsets, w->hd_tokens, w->hd_ctl, seq - 1, w->idle_polls, w->stream);
                w->serving = true;
                ++w->n_launches;
            };
            if (!w->serving) start();
            // the kernel's last store (system-scope release) is the status word; a kernel that left for idleness just before the doorbell
            // rang raises ctl[2] instead: wait for it to be gone, start another
            // A kernel that starts late (a GPU busy with batch kernels or another process, a profiler, more Workers than hardware queues
            // behind another Worker's resident kernel) is not an error: after kWorkerPatience without an answer the kernel is told to
            // leave and the stream is waited for -- it serves a doorbell that already rang on its way out -- and only a failing
            // stream reports VBT_ERR_DEVICE; a sentence that is still unserved then goes through the batch pipeline below.
            constexpr auto kWorkerPatience = std::chrono::seconds(5);
            bool seen = false, gave_up = false;
            auto deadline = std::chrono::steady_clock::time_point{};
            for (uint32_t spins = 0; !seen; ++spins) {
                seen = __atomic_load_n(&w->h_ctl[0], __ATOMIC_ACQUIRE) == seq;
                if (seen) break;
                if (__atomic_load_n(&w->h_ctl[2], __ATOMIC_ACQUIRE) == 1u) {
                    HIPX(hipStreamSynchronize(w->stream));
                    w->serving = false;
                    if (__atomic_load_n(&w->h_ctl[0], __ATOMIC_ACQUIRE) == seq) { seen = true; break; }  // (served on its way out)
                    start();
                } else if ((spins & 0xFFFFu) == 0xFFFFu) {  // (a clock read every 64 k polls)
                    const auto now = std::chrono::steady_clock::now();
                    if (deadline == std::chrono::steady_clock::time_point{}) deadline = now + kWorkerPatience;
                    else if (now > deadline) {
                        __atomic_store_n(&w->h_ctl[6], 1u, __ATOMIC_RELEASE);
                        HIPX(hipStreamSynchronize(w->stream));  // (the only failure that is reported as one)
                        w->h_ctl[6] = 0; w->h_ctl[2] = 0;
                        w->serving = false;
                        seen = __atomic_load_n(&w->h_ctl[0], __ATOMIC_ACQUIRE) == seq;
                        gave_up = !seen;
                        break;
                    }
                }
            }
            if (w->idle_polls == 0 && w->serving) { HIPX(hipStreamSynchronize(w->stream)); w->serving = false; }  // (one launch per call: it has left)
            const uint32_t outcome = gave_up ? 1u : w->h_ctl[3];
            if (gave_up) { w->h_ctl[0] = w->h_ctl[4] = seq; }  // the batch pipeline below serves this sequence number: the next kernel starts behind it
            if (outcome == 0) {
                const uint32_t cnt = w->h_ctl[1];
                if (cnt > len) throw Error(VBT_ERR_INVALID_STATE, "worker: token count exceeds the sentence");
                w->tokens.assign(w->h_tokens, w->h_tokens + cnt);
                ++w->n_fast;
                return;
            }
        }
        w->stop_serving();  // the batch pipeline uses the worker's stream and workspace
        ++w->n_slow;
        const uint64_t offs[2] = {0, len};
        HIPX(hipMemcpy(w->d_text, w->text.data(), len, hipMemcpyHostToDevice));
        HIPX(hipMemcpy(w->d_offsets, offs, 16, hipMemcpyHostToDevice));
        std::vector<uint32_t> off, cnt;
        if (counting) w->ws->reset_connid_counts();  // the device holds the counts of the last lattice only
        run_and_fetch(*w->ws, static_cast<const uint8_t*>(w->d_text), w->d_offsets, 1, len, off, cnt, w->tokens, w->stream);
    });
}

// Sentences this worker served with the single launch / through the batch pipeline.
int vbt_worker_path_stats(const vbt_worker* w, uint64_t* fast, uint64_t* slow) {
    return guarded([&] {
        if (!w) throw Error(VBT_ERR_INVALID_ARGUMENT, "null argument");
        if (fast) *fast = w->n_fast;
        if (slow) *slow = w->n_slow;
    });
}

// The reference's calling pattern, timed in C (tokenize/src/main.rs:78-82, benchmark/src/main.rs:57-61): for every sentence
// reset_sentence -> tokenize -> num_tokens (+ one token() per token), `rounds` passes over the n sentences.
int vbt_worker_loop_benchmark(vbt_worker* w, const uint8_t* text, const uint64_t* offsets, uint64_t n, uint32_t rounds, double* seconds,
                              uint64_t* tokens) {
    if (!w || !offsets || !seconds || !tokens || (!text && n && offsets[n] != offsets[0])) { g_last_error = "null argument"; return VBT_ERR_INVALID_ARGUMENT; }
    uint64_t total = 0;
    const auto t0 = std::chrono::steady_clock::now();
    for (uint32_t r = 0; r < rounds; ++r) {
        for (uint64_t i = 0; i < n; ++i) {
            int rc = vbt_worker_reset_sentence(w, reinterpret_cast<const char*>(text) + offsets[i], offsets[i + 1] - offsets[i]);
            if (rc == VBT_OK) rc = vbt_worker_tokenize(w);
            if (rc != VBT_OK) return rc;
            const uint32_t nt = vbt_worker_num_tokens(w);
            vbt_token t;
            for (uint32_t k = 0; k < nt; ++k) { rc = vbt_worker_token(w, k, &t); if (rc != VBT_OK) return rc; }
            total += nt;
        }
    }
    *seconds = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    *tokens = total;
    return VBT_OK;
}

// Worker::init_connid_counter (worker.rs:77-84): fresh zeroed counters of num_left / num_right entries.
int vbt_worker_init_connid_counter(vbt_worker* w) {
    return guarded([&] {
        if (!w) throw Error(VBT_ERR_INVALID_ARGUMENT, "null argument");
        const Dictionary& d = w->tok->t->dict();
        w->lid_count.assign(d.num_left, 0);
        w->rid_count.assign(d.num_right, 0);
        if (w->ws) {
            HIPX(hipSetDevice(w->tok->t->device()));
            w->ws->enable_connid_counts(true);
            w->ws->reset_connid_counts();
        }
    });
}

// Worker::update_connid_counts (worker.rs:86-93): adds the lattice of the LAST tokenize() call
// (Lattice::add_connid_counts, lattice.rs:170-183).  The reference panics without init_connid_counter: here
// VBT_ERR_INVALID_STATE.  Calling it twice for one tokenize() adds the lattice twice, as the reference does.
int vbt_worker_update_connid_counts(vbt_worker* w) {
    return guarded([&] {
        if (!w) throw Error(VBT_ERR_INVALID_ARGUMENT, "null argument");
        if (w->lid_count.empty()) throw Error(VBT_ERR_INVALID_STATE, "init_connid_counter() has never been called");
        if (!w->ws || w->text.empty()) return;  // no lattice was built for an empty sentence (worker.rs:50-52)
        HIPX(hipSetDevice(w->tok->t->device()));
        std::vector<uint64_t> l(w->lid_count.size()), r(w->rid_count.size());
        w->ws->read_connid_counts(l.data(), r.data(), false);  // kept on the device: a second update adds it again
        for (size_t i = 0; i < l.size(); ++i) w->lid_count[i] += l[i];
        for (size_t i = 0; i < r.size(); ++i) w->rid_count[i] += r[i];
    });
}

int vbt_worker_connid_counts(const vbt_worker* w, uint64_t* lid, uint64_t* rid) {
    return guarded([&] {
        if (!w || !lid || !rid) throw Error(VBT_ERR_INVALID_ARGUMENT, "null argument");
        if (w->lid_count.empty()) throw Error(VBT_ERR_INVALID_STATE, "init_connid_counter() has never been called");
        std::copy(w->lid_count.begin(), w->lid_count.end(), lid);
        std::copy(w->rid_count.begin(), w->rid_count.end(), rid);
    });
}

// ConnIdCounter::compute_probs (mapper.rs:108-146) for one side: drops id 0, sorts by probability descending, then id.
int vbt_connid_probs(const uint64_t* counts, size_t n, uint32_t* ids, double* probs) {
    return guarded([&] {
        if (!counts || !ids || !probs || n == 0) throw Error(VBT_ERR_INVALID_ARGUMENT, "null argument");
        double sum = 0;
        for (size_t i = 0; i < n; ++i) sum += (double)counts[i];
        std::vector<std::pair<uint32_t, double>> v;
        v.reserve(n - 1);
        for (size_t i = 1; i < n; ++i) v.emplace_back((uint32_t)i, (double)counts[i] / sum);
        std::sort(v.begin(), v.end(), [](const auto& a, const auto& b) {
            // partial_cmp(...).unwrap_or(Equal): NaN (0/0) compares equal to everything, then the id decides
            if (a.second > b.second) return true;
            if (a.second < b.second) return false;
            return a.first < b.first;
        });
        for (size_t i = 0; i + 1 < n; ++i) { ids[i] = v[i].first; probs[i] = v[i].second; }
    });
}

int vbt_worker_compute_connid_probs(const vbt_worker* w, uint32_t* lid_ids, double* lid_probs, uint32_t* rid_ids, double* rid_probs) {
    if (!w || w->lid_count.empty()) { g_last_error = "init_connid_counter() has never been called"; return VBT_ERR_INVALID_STATE; }
    int rc = vbt_connid_probs(w->lid_count.data(), w->lid_count.size(), lid_ids, lid_probs);
    if (rc != VBT_OK) return rc;
    return vbt_connid_probs(w->rid_count.data(), w->rid_count.size(), rid_ids, rid_probs);
}

uint32_t vbt_worker_num_tokens(const vbt_worker* w) { return (uint32_t)w->tokens.size(); }

int vbt_worker_token(const vbt_worker* w, uint32_t i, vbt_token* out) {
    return guarded([&] {
        if (i >= w->tokens.size()) throw Error(VBT_ERR_INVALID_ARGUMENT, "token index out of range");
        fill_token(w->tok->t->dict(), reinterpret_cast<const uint8_t*>(w->text.data()), w->tokens[i], out);
    });
}

// Splits n sentences into `parts` contiguous ranges balanced by bytes (vibrato_amd/sharding.py: shard_bounds, the rule the
// multi-process path uses): range k ends at the first sentence that starts at or behind k / parts of the text.
static void shard_bounds(const uint64_t* offs, uint64_t n, uint32_t parts, std::vector<uint64_t>& bounds) {
    bounds.assign(parts + 1, n);
    bounds[0] = 0;
    const uint64_t total = offs[n] - offs[0];
    for (uint32_t r = 1; r < parts; ++r) {
        const uint64_t target = offs[0] + (uint64_t)(((unsigned __int128)total * r) / parts);
        bounds[r] = (uint64_t)(std::lower_bound(offs, offs + n + 1, target) - offs);
        if (bounds[r] > n) bounds[r] = n;
        if (bounds[r] < bounds[r - 1]) bounds[r] = bounds[r - 1];
    }
}

int vbt_tokenize_batch(const vbt_tokenizer* tok_, const uint8_t* text, const uint64_t* offsets, uint64_t n, vbt_batch** out) {
    return guarded([&] {
        vbt_tokenizer* tok = const_cast<vbt_tokenizer*>(tok_);  // the pools are internally synchronised: the handle stays logically const
        if (!tok || !offsets || !out || (!text && n && offsets[n] != offsets[0])) throw Error(VBT_ERR_INVALID_ARGUMENT, "null argument");
        const uint64_t lo = offsets[0];
        for (uint64_t i = 1; i <= n; ++i)
            if (offsets[i] < offsets[i - 1]) throw Error(VBT_ERR_INVALID_ARGUMENT, "offsets must be non-decreasing");
        const uint64_t bytes = offsets[n] - lo;
        if (bytes >= 0xFFFFFFFFull || n >= 0xFFFFFFFFull) throw Error(VBT_ERR_INVALID_ARGUMENT, "batch too large (split it)");
        const uint32_t R = (uint32_t)tok->reps.size();
        // One shard per device (a single-device tokenizer: one shard = the batch).  Every shard gets a pooled workspace + stream
        // of its device; the kernels of all devices run side by side, and every device's DMA engines write its shard of the
        // results into ONE pinned block at the shard's offset: a gather to the caller with no collective and no copy kernel.
        struct Shard { std::unique_ptr<PooledWorkspace> p; uint64_t s0 = 0, s1 = 0, b0 = 0, b1 = 0, tok_base = 0; uint32_t* tail = nullptr; };
        struct Holder {  // blocks and workspaces go back to their pools on every path
            vbt_tokenizer* tok;
            std::unique_ptr<vbt_batch> b;
            std::vector<Shard> sh;
            int alone = 0;                                       // calls in flight on this tokenizer, this one included, when it started
            uint64_t seq = 0;                                    // this call's number
            std::vector<hsa_signal_t> dma;                       // copies into out_blk still in flight (pipelined call)
            std::vector<std::unique_ptr<PooledWorkspace>> pipe;  // the chunk workspaces of a pipelined call (device of reps[0])
            ~Holder() {
                if (tok->calls_in_flight.fetch_sub(1, std::memory_order_relaxed) > 1) tok->last_concurrent.store(tok->call_seq.load(std::memory_order_relaxed), std::memory_order_relaxed);
                (void)Sdma::wait(dma);  // nothing may still be writing into a block that goes back to the pool
                for (auto& p : pipe)
                    if (p) { (void)hipSetDevice(tok->reps[0]->t->device()); (void)hipStreamSynchronize(p->stream); pool_give(tok, *tok->reps[0], std::move(p)); }
                for (size_t k = 0; k < sh.size(); ++k)
                    if (sh[k].p) { (void)hipSetDevice(tok->reps[k]->t->device()); (void)hipStreamSynchronize(sh[k].p->stream); pool_give(tok, *tok->reps[k], std::move(sh[k].p)); }
                if (b) { host_give(tok, std::move(b->in_blk)); host_give(tok, std::move(b->out_blk)); }
            }
        } h{tok, std::make_unique<vbt_batch>(), {}, tok->calls_in_flight.fetch_add(1, std::memory_order_relaxed) + 1, tok->call_seq.fetch_add(1, std::memory_order_relaxed) + 1, {}, {}};
        if (h.alone > 1) tok->last_concurrent.store(h.seq, std::memory_order_relaxed);
        vbt_batch& b = *h.b;
        b.tok = tok;
        b.n = n;
        // the batch's own copy of the input, in pinned memory: [n + 1 rebased offsets][text][8 bytes per shard: totals][per shard: its offsets rebased to the shard]
        // One call, pipelined (single device, results by SDMA): the batch is cut into K chunks that alternate between two chunk-sized
        // workspaces / streams -- the H2D copy and the host's preparation of chunk c + 1 and the D2H copy of chunk c - 1 run under the
        // kernels of chunk c (tokenize/src/main.rs:76-95 is a single-threaded caller: without this, one call serialises copy in ->
        // kernels -> copy out and reaches 20 M sentences/s where the kernels alone do 70).  VBT_H2H_CHUNKS (default 5; 1 = off);
        // chunks of at least 2 MiB.  The result block has to exist before the first chunk's totals do: it is sized from the
        // tokens per KiB the tokenizer's batches produced so far (the first batch runs unpipelined and sets it; a batch that
        // outgrows the estimate is redone unpipelined).
        uint32_t K = 1;
        if (R == 1) {
            static const uint32_t kmax = [] { const char* e = std::getenv("VBT_H2H_CHUNKS"); const int v = e && *e ? std::atoi(e) : 5; return (uint32_t)std::min(std::max(v, 1), 16); }();
            K = (uint32_t)std::min<uint64_t>(kmax, std::max<uint64_t>(1, bytes >> 21));
            if (n < 64ull * K || tok->tok_per_kib.load(std::memory_order_relaxed) == 0) K = 1;
            // several host threads streaming batches overlap each other's copies and kernels already (4 threads: 58-62 M sentences/s
            // whole batches, 51 M with every call cut into chunks): only a lone call is cut
            const uint64_t lc = tok->last_concurrent.load(std::memory_order_relaxed);
            if (h.alone > 1 || (lc && h.seq - lc <= 2)) K = 1;
        }
        const uint32_t parts = std::max(R, K);
        const size_t text_end = (n + 1) * 8 + ((bytes + 7) & ~(uint64_t)7);
        b.in_blk = host_take(tok, text_end + 8 * parts + (parts > 1 ? (n + parts) * 8 : 0) + 24);
        uint64_t* offs = static_cast<uint64_t*>(b.in_blk->p);
        uint8_t* txt = reinterpret_cast<uint8_t*>(offs + n + 1);
        b.offsets = offs;
        b.text = txt;
        uint32_t* tails = reinterpret_cast<uint32_t*>(static_cast<uint8_t*>(b.in_blk->p) + text_end);  // per shard {n_tokens, error flags}
        uint64_t* shard_offs = reinterpret_cast<uint64_t*>(tails + 2 * parts);
        std::vector<uint64_t> bounds;
        shard_bounds(offsets, n, R, bounds);  // (on the caller's offsets: the rule only looks at differences)
        // How the results reach the host (VBT_H2H_OUT): 1 (default) = packed on the device, then copied by the GPU's SDMA engines
        // (Sdma above); 0 = the packing kernel stores them straight into this batch's pinned block.  Either way the total and the
        // error flags come back first, in 8 bytes, so that the pinned block is sized exactly.
        if (tok->out_mode.load(std::memory_order_acquire) < 0) {
            std::lock_guard<std::mutex> g(tok->pool_mu);
            if (tok->out_mode.load(std::memory_order_relaxed) < 0) {
                const char* e = std::getenv("VBT_H2H_OUT");
                int mode = e && *e ? std::atoi(e) : 1;
                if (mode == 1)
                    for (auto& r : tok->reps) {
                        r->sdma = std::make_unique<Sdma>(r->t->device());
                        if (!r->sdma->ok) mode = 0;
                    }
                tok->out_mode.store(mode, std::memory_order_release);  // (behind the Sdma objects: a thread that sees the mode sees them)
            }
        }
        const bool use_sdma = tok->out_mode.load(std::memory_order_acquire) == 1;
        auto note_ratio = [&](uint64_t total) {  // tokens per KiB, rounded up: the largest seen
            if (!bytes) return;
            const uint32_t r = (uint32_t)std::min<uint64_t>(0xFFFFFFu, (total * 1024 + bytes - 1) / bytes + 1);
            uint32_t cur = tok->tok_per_kib.load(std::memory_order_relaxed);
            while (r > cur && !tok->tok_per_kib.compare_exchange_weak(cur, r, std::memory_order_relaxed)) {}
        };
        if (K > 1 && use_sdma) {
            struct Chunk { uint64_t s0, s1, b0, b1, tok_base = 0; uint32_t* tail; size_t first_sig = 0, n_sigs = 0; };
            // chunk borders by bytes, at sentence borders; the first and the last chunk half as long as the others: the head of the call
            // (copying the first chunk together, its H2D) and its tail (the last chunk's D2H) are what nothing overlaps
            std::vector<uint64_t> cb(K + 1, n);
            cb[0] = 0;
            {
                const uint32_t wt = K >= 3 ? 2 * (K - 1) : K;
                uint32_t acc = 0;
                for (uint32_t c = 1; c < K; ++c) {
                    acc += K >= 3 ? (c == 1 ? 1u : 2u) : 1u;
                    const uint64_t target = lo + (uint64_t)(((unsigned __int128)bytes * acc) / wt);
                    cb[c] = std::min<uint64_t>(n, (uint64_t)(std::lower_bound(offsets, offsets + n + 1, target) - offsets));
                    if (cb[c] < cb[c - 1]) cb[c] = cb[c - 1];
                }
            }
            std::vector<Chunk> ch(K);
            uint64_t max_s = 0, max_b = 0;
            for (uint32_t c = 0; c < K; ++c) {
                ch[c].s0 = cb[c]; ch[c].s1 = cb[c + 1];
                ch[c].b0 = offsets[ch[c].s0] - lo; ch[c].b1 = offsets[ch[c].s1] - lo;
                ch[c].tail = tails + 2 * c;
                ch[c].tail[0] = ch[c].tail[1] = 0;
                max_s = std::max(max_s, ch[c].s1 - ch[c].s0); max_b = std::max(max_b, ch[c].b1 - ch[c].b0);
            }
            Replica& rep = *tok->reps[0];
            HIPX(hipSetDevice(rep.t->device()));
            // three chunk workspaces in rotation: chunk c + 1 is enqueued while c runs and c - 1 is on its way out (with two, the
            // enqueue of chunk c waited for the copy of c - 2 to land and the GPU idled in between: traced)
            const uint32_t W = std::min<uint32_t>(3, K);
            for (uint32_t w = 0; w < W; ++w) h.pipe.push_back(pool_take(tok, rep, max_s, max_b));
            // the result block, from the estimate: 1/8 of head room + a token per sentence
            const uint64_t cap_tokens = std::min<uint64_t>(bytes, (bytes * tok->tok_per_kib.load(std::memory_order_relaxed) / 1024) * 9 / 8 + n + 1024);
            b.out_blk = host_take(tok, n * 8 + (size_t)cap_tokens * sizeof(vbt_token_rec) + 16);
            uint32_t* o = static_cast<uint32_t*>(b.out_blk->p);
            vbt_token_rec* otok = reinterpret_cast<vbt_token_rec*>(o + 2 * n);
            uint64_t total = 0;
            uint32_t error_flags = 0;
            bool overflow = false;
            std::vector<std::pair<size_t, size_t>> sig_range(K, {0, 0});
            // chunk c's kernels are done: its totals are in; its results start their way to the host on the DMA engines
            // The kernels of all chunks go to ONE stream, in chunk order, every chunk followed by an event: how many hardware queues the
            // process has (HIP's default is four, shared by every stream of the application) and which streams share one is then
            // irrelevant -- on streams of their own the chunks of a pipelined call ran 25 % slower than an unpipelined call inside a
            // process that had initialised HIP before this library.  VBT_H2H_ONE_STREAM=0: a stream per workspace (A/B).
            static const bool one_stream = [] { const char* e = std::getenv("VBT_H2H_ONE_STREAM"); return !(e && *e == '0'); }();
            auto stream_of = [&](uint32_t c) { return one_stream ? h.pipe[0]->stream : h.pipe[c % W]->stream; };
            auto harvest = [&](uint32_t c) {
                PooledWorkspace& p = *h.pipe[c % W];
                HIPX(hipEventSynchronize(p.done));
                Chunk& q = ch[c];
                error_flags |= q.tail[1];
                q.tok_base = total;
                if (q.tail[1] & kErrFatal) return;  // (a rejected batch: nothing to copy)
                if (total + q.tail[0] > cap_tokens) { overflow = true; return; }
                total += q.tail[0];
                const uint64_t ns = q.s1 - q.s0;
                const void* src[3] = {p.ws->d_tok_off, p.ws->d_tok_cnt, p.ws->d_tokens};
                void* dst[3] = {o + q.s0, o + n + q.s0, otok + q.tok_base};
                const size_t len[3] = {(size_t)ns * 4, (size_t)ns * 4, (size_t)q.tail[0] * sizeof(vbt_token_rec)};
                sig_range[c].first = h.dma.size();
                if (!rep.sdma->issue(src, dst, len, 3, h.dma)) throw Error(VBT_ERR_DEVICE, "device -> host copy of the results failed (hsa_amd_memory_async_copy)");
                sig_range[c].second = h.dma.size();
            };
            // chunk c's copies have landed (before its workspace is used again, and at the end): rebase its token offsets to the batch
            auto land = [&](uint32_t c) {
                bool good = true;
                for (size_t i = sig_range[c].first; i < sig_range[c].second; ++i) {
                    hsa_signal_t sig = h.dma[i];
                    if (!sig.handle) continue;
                    while (hsa_signal_wait_scacquire(sig, HSA_SIGNAL_CONDITION_LT, 1, UINT64_MAX, HSA_WAIT_STATE_ACTIVE) >= 1) {}
                    if (hsa_signal_load_relaxed(sig) < 0) good = false;
                    (void)hsa_signal_destroy(sig);
                    h.dma[i].handle = 0;
                }
                if (!good) throw Error(VBT_ERR_DEVICE, "device -> host copy of the results failed (hsa_amd_memory_async_copy)");
                if (ch[c].tok_base)
                    for (uint64_t i = ch[c].s0; i < ch[c].s1; ++i) o[i] += (uint32_t)ch[c].tok_base;
            };
            static const bool trace = std::getenv("VBT_H2H_TRACE") != nullptr;  // developer aid: where a pipelined call's time goes (us since its start)
            const auto t_start = std::chrono::steady_clock::now();
            auto stamp = [&](const char* what, uint32_t c) {
                if (trace) std::fprintf(stderr, "[vbt h2h] %-8s chunk %u at %7.1f us\n", what, c, std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t_start).count());
            };
            auto prep_enqueue = [&](uint32_t c) {
                Chunk& q = ch[c];
                const uint64_t ns = q.s1 - q.s0, nb = q.b1 - q.b0;
                // the batch's own copy of this chunk's input, and its offsets rebased to the chunk
                for (uint64_t i = q.s0; i < q.s1; ++i) offs[i] = offsets[i] - lo;
                if (c + 1 == K) offs[n] = bytes;
                if (nb) std::memcpy(txt + q.b0, text + lo + q.b0, nb);
                uint64_t* so = shard_offs + q.s0 + c;
                for (uint64_t i = 0; i <= ns; ++i) so[i] = offsets[q.s0 + i] - lo - q.b0;
                stamp("prepared", c);
                if (c >= W) { land(c - W); stamp("landed", c - W); }  // the workspace's previous results are out
                PooledWorkspace& p = *h.pipe[c % W];
                if (ns) {
                    // the way in on the DMA engines too (Sdma::issue_h2d: why); the kernels are enqueued once the chunk's input has landed --
                    // the host is W chunks ahead of the device, this wait costs the call nothing
                    std::vector<hsa_signal_t> in;
                    bool ok = rep.sdma->issue_h2d(txt + q.b0, p.d_text, nb, in);
                    ok = rep.sdma->issue_h2d(so, p.d_off, (ns + 1) * 8, in) && ok;
                    ok = Sdma::wait(in) && ok;
                    if (!ok) throw Error(VBT_ERR_DEVICE, "host -> device copy of a chunk failed (hsa_amd_memory_async_copy)");
                    p.ws->run(static_cast<const uint8_t*>(p.d_text), p.d_off, ns, nb, stream_of(c), /*defer_pack=*/false);
                    HIPX(hipMemcpyAsync(q.tail, p.ws->d_ctrl, 8, hipMemcpyDeviceToHost, stream_of(c)));
                }
                HIPX(hipEventRecord(p.done, stream_of(c)));
                stamp("enqueued", c);
            };
            // The host runs ahead of the device by as many chunks as there are workspaces: it prepares and enqueues chunk c + W - 1 before
            // it blocks for the totals of chunk c (blocking first left the GPU idle while the next chunk was being copied together).
            for (uint32_t enq = 0, har = 0; har < K && !overflow; ++har) {
                while (enq < K && enq < har + W) prep_enqueue(enq++);
                harvest(har);
                stamp("harvest", har);
            }
            if (!overflow) {
                for (uint32_t c = K - W; c < K; ++c) { land(c); stamp("landed", c); }
                if (error_flags & kErrUtf8) {
                    for (uint64_t i = 0; i < n; ++i)
                        if (!valid_utf8(txt + offs[i], offs[i + 1] - offs[i])) throw Error(VBT_ERR_UTF8, "sentence " + std::to_string(i) + " is not valid UTF-8");
                }
                check_device_errors(error_flags);
                b.n_tokens = total;
                b.tok_off = o; b.tok_cnt = o + n; b.tokens = otok;
                note_ratio(total);
                for (auto& p : h.pipe) pool_give(tok, rep, std::move(p));
                h.pipe.clear();
                *out = h.b.release();
                return;
            }
            // the estimate was too low for this text: drain, give everything back and run the batch unpipelined (exact sizing)
            (void)Sdma::wait(h.dma);
            for (auto& p : h.pipe) { HIPX(hipStreamSynchronize(p->stream)); pool_give(tok, rep, std::move(p)); }
            h.pipe.clear();
            host_give(tok, std::move(b.out_blk));
            for (uint32_t c = 0; c < K; ++c) ch[c].tail[0] = ch[c].tail[1] = 0;
        }
        h.sh.resize(R);
        // Every device is driven by ITS OWN host thread (a single-device tokenizer: the calling thread): the thread rebases its shard's
        // offsets, copies its shard of the text into the pinned block, enqueues the H2D copies and the launch sequence on its device
        // and waits for the 8 bytes of totals -- so eight devices cost one shard's host work, not eight in a row (round-4 review).
        auto in_shard_threads = [&](auto&& body) {
            if (R == 1) { body(0u); return; }
            std::vector<std::exception_ptr> errs(R);
            std::vector<std::thread> th;
            auto run = [&](uint32_t k) { try { body(k); } catch (...) { errs[k] = std::current_exception(); } };
            // the thread of device k runs on the CPUs of that device's NUMA node (VBT_NUMA_PIN=0: wherever the OS puts it); the calling
            // thread, which takes device 0, stays where its owner put it
            static const bool pin = [] { const char* e = std::getenv("VBT_NUMA_PIN"); return !(e && *e == '0'); }();
            auto run_pinned = [&](uint32_t k) {
                const std::vector<int>& cpus = tok->reps[k]->near_cpus;
                if (pin && !cpus.empty()) {
                    cpu_set_t set;
                    CPU_ZERO(&set);
                    for (int c : cpus) CPU_SET(c, &set);
                    (void)pthread_setaffinity_np(pthread_self(), sizeof(set), &set);
                }
                run(k);
            };
            uint32_t started = 1;
            try {
                th.reserve(R);
                for (; started < R; ++started) th.emplace_back(run_pinned, started);
            } catch (const std::exception&) {}  // (out of threads: the rest runs here)
            run(0);
            for (uint32_t k = started; k < R; ++k) run(k);
            for (auto& t : th) t.join();
            for (auto& e : errs) if (e) std::rethrow_exception(e);
        };
        in_shard_threads([&](uint32_t k) {
            Shard& s = h.sh[k];
            s.s0 = bounds[k]; s.s1 = bounds[k + 1];
            s.b0 = offsets[s.s0] - lo; s.b1 = offsets[s.s1] - lo;
            s.tail = tails + 2 * k;
            s.tail[0] = s.tail[1] = 0;
            const uint64_t ns = s.s1 - s.s0, nb = s.b1 - s.b0;
            // this shard's part of the batch's own copy of the input (the last shard also writes offs[n])
            for (uint64_t i = s.s0; i < s.s1; ++i) offs[i] = offsets[i] - lo;
            if (k + 1 == R) offs[n] = bytes;
            if (nb) std::memcpy(txt + s.b0, text + lo + s.b0, nb);
            if (ns == 0) return;
            const uint64_t* so = offs;  // a single shard reads the batch's offsets as they are
            if (R > 1) {
                uint64_t* q = shard_offs + s.s0 + k;
                for (uint64_t i = 0; i <= ns; ++i) q[i] = offsets[s.s0 + i] - lo - s.b0;
                so = q;
            }
            HIPX(hipSetDevice(tok->reps[k]->t->device()));
            s.p = pool_take(tok, *tok->reps[k], ns, nb);
            PooledWorkspace& p = *s.p;
            if (nb) HIPX(hipMemcpyAsync(p.d_text, txt + s.b0, nb, hipMemcpyHostToDevice, p.stream));
            HIPX(hipMemcpyAsync(p.d_off, so, (ns + 1) * 8, hipMemcpyHostToDevice, p.stream));
            p.ws->run(static_cast<const uint8_t*>(p.d_text), p.d_off, ns, nb, p.stream, /*defer_pack=*/!use_sdma);
            HIPX(hipMemcpyAsync(s.tail, p.ws->d_ctrl, 8, hipMemcpyDeviceToHost, p.stream));
            HIPX(hipStreamSynchronize(p.stream));
        });
        uint32_t error_flags = 0;
        uint64_t total = 0;
        for (uint32_t k = 0; k < R; ++k) {
            Shard& s = h.sh[k];
            if (!s.p) continue;
            error_flags |= s.tail[1];
            s.tok_base = total;
            total += s.tail[0];
        }
        if (error_flags & kErrUtf8) {
            // every sentence must be a Rust `str` (the reference's callers pass &str; its CLI fails on invalid input lines).  The
            // device's first kernel validates the text; only this error path walks it again on the host to name the sentence.
            for (uint64_t i = 0; i < n; ++i)
                if (!valid_utf8(txt + offs[i], offs[i + 1] - offs[i])) throw Error(VBT_ERR_UTF8, "sentence " + std::to_string(i) + " is not valid UTF-8");
        }
        check_device_errors(error_flags);
        if (total >= 0xFFFFFFFFull) throw Error(VBT_ERR_INVALID_ARGUMENT, "batch too large (split it)");
        b.n_tokens = total;
        note_ratio(total);
        b.out_blk = host_take(tok, n * 8 + (size_t)b.n_tokens * sizeof(vbt_token_rec) + 16);
        uint32_t* o = static_cast<uint32_t*>(b.out_blk->p);
        vbt_token_rec* otok = reinterpret_cast<vbt_token_rec*>(o + 2 * n);
        b.tok_off = o;
        b.tok_cnt = o + n;
        b.tokens = otok;
        // the results: every device's own thread has its DMA engines (or its packing kernel) write its shard into the one block at
        // the shard's place, waits for them, rebases the shard's token offsets to the batch and hands its workspace back
        in_shard_threads([&](uint32_t k) {
            Shard& s = h.sh[k];
            if (!s.p) return;
            const uint64_t ns = s.s1 - s.s0;
            PooledWorkspace& p = *s.p;
            if (use_sdma) {
                const void* src[3] = {p.ws->d_tok_off, p.ws->d_tok_cnt, p.ws->d_tokens};
                void* dst[3] = {o + s.s0, o + n + s.s0, otok + s.tok_base};
                const size_t len[3] = {(size_t)ns * 4, (size_t)ns * 4, (size_t)s.tail[0] * sizeof(vbt_token_rec)};
                std::vector<hsa_signal_t> sigs;
                bool ok = tok->reps[k]->sdma->issue(src, dst, len, 3, sigs);
                ok = Sdma::wait(sigs) && ok;
                if (!ok) throw Error(VBT_ERR_DEVICE, "device -> host copy of the results failed (hsa_amd_memory_async_copy)");
            } else {
                HIPX(hipSetDevice(tok->reps[k]->t->device()));
                void* dev = nullptr;
                HIPX(hipHostGetDevicePointer(&dev, o, 0));
                uint32_t* od = static_cast<uint32_t*>(dev);
                p.ws->pack_to(reinterpret_cast<vbt_token_rec*>(od + 2 * n) + s.tok_base, od + s.s0, od + n + s.s0, p.stream);
                HIPX(hipStreamSynchronize(p.stream));
            }
            if (s.tok_base)  // a shard's token offsets start at 0: rebase them to the batch
                for (uint64_t i = s.s0; i < s.s1; ++i) o[i] += (uint32_t)s.tok_base;
            pool_give(tok, *tok->reps[k], std::move(s.p));
        });
        *out = h.b.release();
    });
}

// Releases every idle pooled workspace and pinned block now (they are re-created on demand).
int vbt_tokenizer_trim_pool(const vbt_tokenizer* tok_) {
    return guarded([&] {
        vbt_tokenizer* tok = const_cast<vbt_tokenizer*>(tok_);
        if (!tok) throw Error(VBT_ERR_INVALID_ARGUMENT, "null argument");
        std::vector<std::vector<std::unique_ptr<PooledWorkspace>>> ws(tok->reps.size());
        std::vector<std::unique_ptr<PinnedBlock>> blocks;
        {
            std::lock_guard<std::mutex> g(tok->pool_mu);
            for (size_t k = 0; k < tok->reps.size(); ++k) ws[k].swap(tok->reps[k]->pool);
            blocks.swap(tok->host_pool);
        }
        for (size_t k = 0; k < ws.size(); ++k) { HIPX(hipSetDevice(tok->reps[k]->t->device())); ws[k].clear(); }
        out_cache_trim();  // the formatter's idle output buffers (process-wide)
    });
}

// Workspaces created / reused by vbt_tokenize_batch so far (steady state: created stops growing).
int vbt_tokenizer_pool_stats(const vbt_tokenizer* tok_, uint64_t* created, uint64_t* reused, uint64_t* idle) {
    return guarded([&] {
        vbt_tokenizer* tok = const_cast<vbt_tokenizer*>(tok_);
        if (!tok) throw Error(VBT_ERR_INVALID_ARGUMENT, "null argument");
        std::lock_guard<std::mutex> g(tok->pool_mu);
        if (created) *created = tok->pool_created;
        if (reused) *reused = tok->pool_reused;
        if (idle) { *idle = 0; for (const auto& r : tok->reps) *idle += r->pool.size(); }
    });
}

void vbt_batch_free(vbt_batch* b) {
    if (!b) return;
    host_give(b->tok, std::move(b->in_blk));  // the tokenizer outlives its batches (as it outlives its workers)
    host_give(b->tok, std::move(b->out_blk));
    delete b;
}
uint64_t vbt_batch_num_sentences(const vbt_batch* b) { return b->n; }
uint64_t vbt_batch_total_tokens(const vbt_batch* b) { return b->n_tokens; }
uint32_t vbt_batch_num_tokens(const vbt_batch* b, uint64_t s) { return s < b->n ? b->tok_cnt[s] : 0; }
const vbt_token_rec* vbt_batch_records(const vbt_batch* b, uint64_t s) {
    return s < b->n && b->tok_cnt[s] ? &b->tokens[b->tok_off[s]] : nullptr;
}

int vbt_batch_arrays(const vbt_batch* b, const vbt_token_rec** tokens, const uint32_t** tok_off, const uint32_t** tok_cnt) {
    if (tokens) *tokens = b->tokens;
    if (tok_off) *tok_off = b->tok_off;
    if (tok_cnt) *tok_cnt = b->tok_cnt;
    return VBT_OK;
}

int vbt_batch_token(const vbt_batch* b, uint64_t s, uint32_t i, vbt_token* out) {
    return guarded([&] {
        if (s >= b->n || i >= b->tok_cnt[s]) throw Error(VBT_ERR_INVALID_ARGUMENT, "token index out of range");
        fill_token(b->tok->t->dict(), b->text + b->offsets[s], b->tokens[b->tok_off[s] + i], out);
    });
}

// tokenize's output stage (tokenize/src/main.rs:83-127) for a whole batch.  The text of 100 k sentences is ~230 MB in mecab mode:
// a memcpy-shaped job, done in two parallel passes over chunks of sentences balanced by tokens -- exact byte sizes first (a
// prefix over the chunks gives every chunk its place in the one output buffer), then every thread renders its chunk in place.
// No intermediate strings, integers formatted by hand.  VBT_FORMAT_THREADS (default: the host's cores, at most 32; small
// batches use fewer: one thread per 2 MB of output).
namespace {
inline size_t dec_len(uint32_t v) { size_t n = 1; while (v >= 10) { v /= 10; ++n; } return n; }
inline size_t dec_len_i(int32_t v) { return v < 0 ? 1 + dec_len(0u - (uint32_t)v) : dec_len((uint32_t)v); }
inline char* put_u(char* p, uint32_t v) {
    char tmp[10];
    int k = 0;
    do { tmp[k++] = (char)('0' + v % 10); v /= 10; } while (v);
    while (k) *p++ = tmp[--k];
    return p;
}
inline char* put_i(char* p, int32_t v) {
    if (v < 0) { *p++ = '-'; return put_u(p, 0u - (uint32_t)v); }
    return put_u(p, (uint32_t)v);
}
inline char* put_s(char* p, const char* s, size_t n) { std::memcpy(p, s, n); return p + n; }
const char* const kLexNames[3] = {"System", "User", "Unknown"};  // {:?} of LexType
const size_t kLexNameLen[3] = {6, 4, 7};

size_t format_size(const vbt_token& t, int mode) {
    if (mode == VBT_FORMAT_WAKATI) return t.surface_len;
    size_t n = t.surface_len + 1 + t.feature_len + 1;
    if (mode == VBT_FORMAT_DETAIL)  // "\tlex_type=%s\tleft_id=%u\tright_id=%u\tword_cost=%d\ttotal_cost=%d"
        n += 10 + kLexNameLen[t.lex_type] + 9 + dec_len(t.left_id) + 10 + dec_len(t.right_id) + 11 + dec_len_i(t.word_cost) + 12 + dec_len_i(t.total_cost);
    return n;
}
char* format_token(char* p, const vbt_token& t, int mode) {
    p = put_s(p, t.surface, t.surface_len);
    if (mode == VBT_FORMAT_WAKATI) return p;
    *p++ = '\t';
    p = put_s(p, t.feature, t.feature_len);
    if (mode == VBT_FORMAT_DETAIL) {  // tokenize/src/main.rs:108-123
        p = put_s(p, "\tlex_type=", 10); p = put_s(p, kLexNames[t.lex_type], kLexNameLen[t.lex_type]);
        p = put_s(p, "\tleft_id=", 9); p = put_u(p, t.left_id);
        p = put_s(p, "\tright_id=", 10); p = put_u(p, t.right_id);
        p = put_s(p, "\tword_cost=", 11); p = put_i(p, t.word_cost);
        p = put_s(p, "\ttotal_cost=", 12); p = put_i(p, t.total_cost);
    }
    *p++ = '\n';
    return p;
}
}  // namespace

int vbt_batch_format(const vbt_batch* b, int mode, char** out, size_t* len) {
    return guarded([&] {
        if (!b || !out || !len) throw Error(VBT_ERR_INVALID_ARGUMENT, "null argument");
        if (mode < VBT_FORMAT_MECAB || mode > VBT_FORMAT_DETAIL) throw Error(VBT_ERR_INVALID_ARGUMENT, "mode: unknown output mode");
        const Dictionary& d = b->tok->t->dict();
        const uint64_t n = b->n;
        // The feature strings, flat: one blob per lexicon + a u32 offset per word id (3.5 MB of offsets for unidic: they stay in the
        // host's caches, where 877 k separately allocated std::strings cost a cache miss per token for the length alone).
        vbt_tokenizer* tk = b->tok;
        std::call_once(tk->flat_once, [&] {
            auto flatten = [](const std::vector<std::string>& f, vbt_tokenizer::FlatFeatures& o) {
                size_t total = 0;
                for (const auto& x : f) total += x.size();
                if (total >= 0xFFFFFFFFull) throw Error(VBT_ERR_UNSUPPORTED, "format: more than 4 GiB of feature strings");
                o.off.resize(f.size() + 1);
                o.blob.resize(total);
                size_t at = 0;
                for (size_t i = 0; i < f.size(); ++i) { o.off[i] = (uint32_t)at; std::memcpy(o.blob.data() + at, f[i].data(), f[i].size()); at += f[i].size(); }
                o.off[f.size()] = (uint32_t)at;
            };
            flatten(d.system.features, tk->flat[VBT_LEX_SYSTEM]);
            if (d.has_user) flatten(d.user.features, tk->flat[VBT_LEX_USER]); else tk->flat[VBT_LEX_USER].off.assign(1, 0);
            flatten(d.unk_features, tk->flat[VBT_LEX_UNKNOWN]);
        });
        const vbt_tokenizer::FlatFeatures* flat = tk->flat;
        // chunks of sentences balanced by tokens (tok_off is non-decreasing in sentence order)
        unsigned want = std::thread::hardware_concurrency();
        if (want == 0) want = 1;
        if (want > 64) want = 64;
        if (const char* e = std::getenv("VBT_FORMAT_THREADS")) { const int v = std::atoi(e); if (v > 0) want = (unsigned)std::min(v, 256); }
        const uint64_t est = b->n_tokens * 40 + n * 4;
        unsigned T = (unsigned)std::max<uint64_t>(1, std::min<uint64_t>(want, est / (1u << 20) + 1));
        // persistent workers (HostPool): the number granted (a host out of threads grants fewer) decides how many chunks there are,
        // so the barrier between the two passes below counts exactly the threads that take part
        HostPool& pool = HostPool::get();
        T = pool.begin(T);
        struct PoolGuard { HostPool& p; bool ran = false; ~PoolGuard() { if (!ran) p.cancel(); } } pool_guard{pool};
        std::function<void(unsigned)> chunk_body;
        std::vector<uint64_t> first(T + 1, n);
        first[0] = 0;
        for (unsigned k = 1; k < T; ++k) {
            const uint32_t target = (uint32_t)(b->n_tokens * k / T);
            first[k] = (uint64_t)(std::lower_bound(b->tok_off, b->tok_off + n, target) - b->tok_off);
            if (first[k] < first[k - 1]) first[k] = first[k - 1];
        }
        // feature string of a record (mecab mode): two reads of the cached offset array; a word id outside the dictionary is an error
        auto feature_of = [&](const vbt_token_rec& r, const char*& p, size_t& l) {
            const uint32_t lex = r.word_idx >> 30, wid = r.word_idx & 0x3FFFFFFFu;
            if (lex > VBT_LEX_UNKNOWN || (size_t)wid + 1 >= flat[lex].off.size()) throw std::runtime_error("format: a token's word id lies outside its lexicon");
            const uint32_t a = flat[lex].off[wid];
            p = flat[lex].blob.data() + a;
            l = flat[lex].off[wid + 1] - a;
        };
        auto sentence_size = [&](uint64_t si) {
            const uint32_t nt = b->tok_cnt[si];
            const vbt_token_rec* recs = b->tokens + b->tok_off[si];
            size_t sz = mode == VBT_FORMAT_WAKATI ? (nt ? nt - 1 : 0) + 1 : 4;  // separators + '\n' | "EOS\n" (tokenize/src/main.rs:91,96-103)
            if (mode == VBT_FORMAT_DETAIL) {
                const uint8_t* sent = b->text + b->offsets[si];
                vbt_token t;
                for (uint32_t i = 0; i < nt; ++i) { fill_token(d, sent, recs[i], &t); sz += format_size(t, mode); }
            } else if (mode == VBT_FORMAT_WAKATI) {
                for (uint32_t i = 0; i < nt; ++i) sz += recs[i].end_byte - recs[i].start_byte;
            } else {
                const char* fp; size_t fl;
                for (uint32_t i = 0; i < nt; ++i) { feature_of(recs[i], fp, fl); sz += (recs[i].end_byte - recs[i].start_byte) + fl + 2; }
            }
            return sz;
        };
        auto render = [&](uint64_t si, char* p) {
            const uint8_t* sent = b->text + b->offsets[si];
            const uint32_t nt = b->tok_cnt[si];
            const vbt_token_rec* recs = b->tokens + b->tok_off[si];
            if (mode == VBT_FORMAT_DETAIL) {
                vbt_token t;
                for (uint32_t i = 0; i < nt; ++i) { fill_token(d, sent, recs[i], &t); p = format_token(p, t, mode); }
            } else if (mode == VBT_FORMAT_WAKATI) {
                for (uint32_t i = 0; i < nt; ++i) {
                    if (i) *p++ = ' ';
                    p = put_s(p, reinterpret_cast<const char*>(sent) + recs[i].start_byte, recs[i].end_byte - recs[i].start_byte);
                }
            } else {  // surface \t feature \n (tokenize/src/main.rs:83-91)
                const char* fp; size_t fl;
                for (uint32_t i = 0; i < nt; ++i) {
                    feature_of(recs[i], fp, fl);
                    p = put_s(p, reinterpret_cast<const char*>(sent) + recs[i].start_byte, recs[i].end_byte - recs[i].start_byte);
                    *p++ = '\t';
                    p = put_s(p, fp, fl);
                    *p++ = '\n';
                }
            }
            if (mode == VBT_FORMAT_WAKATI) *p++ = '\n';
            else p = put_s(p, "EOS\n", 4);
            return p;
        };
        std::vector<size_t> chunk_bytes(T, 0);
        std::string failure;
        std::mutex fail_mu;
        // Both passes on the one set of threads: sizes, then -- behind a barrier at which the last arrival takes the prefix over the
        // chunks and allocates -- every thread renders its own chunk in place (and first-touches its part of the output).
        size_t total = 0;
        std::vector<size_t> at(T + 1, 0);
        char* buf = nullptr;
        std::mutex bar_mu;
        std::condition_variable bar_cv;
        unsigned arrived = 0;
        bool ready = false, alloc_failed = false;
        auto fail_with = [&](const char* what) { std::lock_guard<std::mutex> g(fail_mu); failure = what; };
        chunk_body = [&](unsigned k) {
            size_t sz = 0;
            bool ok = true;
            try { for (uint64_t si = first[k]; si < first[k + 1]; ++si) sz += sentence_size(si); }
            catch (const std::exception& e) { ok = false; fail_with(e.what()); }
            {
                std::unique_lock<std::mutex> g(bar_mu);
                chunk_bytes[k] = sz;
                if (++arrived == T) {
                    for (unsigned q = 0; q < T; ++q) { at[q] = total; total += chunk_bytes[q]; }
                    at[T] = total;
                    bool failed;
                    { std::lock_guard<std::mutex> f(fail_mu); failed = !failure.empty(); }
                    if (!failed) buf = static_cast<char*>(out_alloc(total + 1));
                    alloc_failed = !failed && !buf;
                    ready = true;
                    bar_cv.notify_all();
                } else bar_cv.wait(g, [&] { return ready; });
            }
            if (!ok || !buf) return;
            try {
                char* p = buf + at[k];
                for (uint64_t si = first[k]; si < first[k + 1]; ++si) p = render(si, p);
                if (p != buf + at[k + 1]) fail_with("format: size pass and render pass disagree");
            } catch (const std::exception& e) { fail_with(e.what()); }
        };
        pool_guard.ran = true;
        pool.run(T, chunk_body);
        if (!failure.empty()) { out_free(buf); throw Error(VBT_ERR_INVALID_STATE, failure); }
        if (alloc_failed) throw std::bad_alloc();
        buf[total] = 0;
        *out = buf;
        *len = total;
    });
}

void vbt_free(void* p) { out_free(p); }

int vbt_workspace_new(const vbt_tokenizer* tok, uint64_t max_sentences, uint64_t max_bytes, vbt_workspace** out) {
    return guarded([&] {
        if (!tok || !out) throw Error(VBT_ERR_INVALID_ARGUMENT, "null argument");
        *out = new vbt_workspace{std::make_unique<Workspace>(*tok->t, max_sentences, max_bytes)};
    });
}

void vbt_workspace_free(vbt_workspace* ws) { delete ws; }

int vbt_tokenize_batch_device(vbt_workspace* ws, const uint8_t* d_text, const uint64_t* d_offsets, uint64_t n, uint64_t total_bytes,
                              void* hip_stream) {
    return guarded([&] { ws->w->run(d_text, d_offsets, n, total_bytes, hip_stream); });
}

int vbt_workspace_results(const vbt_workspace* ws, const vbt_token_rec** d_tokens, const uint32_t** d_tok_off, const uint32_t** d_tok_cnt,
                          const uint32_t** d_total) {
    return guarded([&] {
        const Workspace& w = *ws->w;
        char* slot = static_cast<char*>(w.packed_slot);  // (with a packed output slot set: where inside it the arrays are)
        if (d_tokens) *d_tokens = slot ? reinterpret_cast<const vbt_token_rec*>(slot + 32 + 8 * w.packed_max_s) : w.d_tokens;
        if (d_tok_off) *d_tok_off = slot ? reinterpret_cast<const uint32_t*>(slot + 32) : w.d_tok_off;
        if (d_tok_cnt) *d_tok_cnt = slot ? reinterpret_cast<const uint32_t*>(slot + 32 + 4 * w.packed_max_s) : w.d_tok_cnt;
        if (d_total) *d_total = w.d_ctrl;
    });
}

int vbt_workspace_set_packed_output(vbt_workspace* ws, void* d_slot, uint64_t slot_bytes, uint64_t max_sentences) {
    return guarded([&] {
        if (!ws) throw Error(VBT_ERR_INVALID_ARGUMENT, "null argument");
        ws->w->set_packed_output(d_slot, slot_bytes, max_sentences);
    });
}

int vbt_workspace_set_timing(vbt_workspace* ws, int enabled) {
    ws->w->timing = enabled != 0;
    return VBT_OK;
}

int vbt_workspace_count_connids(vbt_workspace* ws, int enabled) {
    return guarded([&] { ws->w->enable_connid_counts(enabled != 0); });
}

int vbt_workspace_connid_counts(vbt_workspace* ws, uint64_t* lid, uint64_t* rid, int reset) {
    return guarded([&] { ws->w->read_connid_counts(lid, rid, reset != 0); });
}

int vbt_workspace_profile(vbt_workspace* ws, uint64_t out[12], int reset) {
    return guarded([&] { ws->w->read_profile(out, reset != 0); });
}

#ifdef VBT_DEBUG_API
// Developer aid (debug build variants only, not part of the ABI): device addresses of a workspace's intermediate arrays.
__attribute__((visibility("default"))) int vbt_debug_ptrs(vbt_workspace* ws, uint64_t out[8]) {
    const BatchArgs& p = ws->w->pipe;
    out[0] = (uint64_t)(uintptr_t)p.s_hdr; out[1] = (uint64_t)(uintptr_t)p.g_pc; out[2] = (uint64_t)(uintptr_t)p.g_cand;
    out[3] = (uint64_t)(uintptr_t)p.g_hits; out[4] = (uint64_t)(uintptr_t)p.g_c2b; out[5] = p.node_factor; out[6] = kSentenceSlack; out[7] = 0;
    return VBT_OK;
}
#endif

int vbt_workspace_stats(vbt_workspace* ws, vbt_call_stats* out) {
    return guarded([&] { ws->w->stats(out); });
}

}  // extern "C"
