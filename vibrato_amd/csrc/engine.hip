// Host side of the device engine of the MI355X tokenizer: the dictionary image in HBM (Tokenizer), the batch workspace and the
// launch sequence of one batch (Workspace).  The kernels live in gen.hip (validate_batch, gen_candidates, gen_candidates_large,
// build_lists), lattice.hip (lattice_lds, tokenize_serve), fused.hip (the single-kernel fallback) and pack.hip (compact_tokens,
// expand_connector), behind the launchers of kernels.hpp; device_common.hpp / gen_device.hpp hold what they share.  DESIGN.md
// section 3 is the map:
//
// Pipeline (default): validate_batch (device-side input contract) -> gen_candidates (one wavefront per sentence: UTF-8 decode, char
// categories and groupable runs -- Sentence::compile, sentence.rs:34-71 -- and candidate generation by double-array common-prefix
// search + unknown-word rules -- Tokenizer::add_lattice_edges tokenizer.rs:141-199, UnkHandler::gen_unk_words unknown.rs:69-137)
// -> build_lists -> gen_candidates_large (gen_long: one WORKGROUP per sentence that outgrew the bulk generator's LDS) ->
// lattice_lds (ONE 10 KiB tier; one wavefront per sentence: the position sweep with per-node min-cost search over the connection
// matrix -- build_lattice_inner tokenizer.rs:94-139, Lattice::insert_node / search_min_node lattice.rs:103-151, insert_eos 85-101
// -- and the back-trace, append_top_nodes lattice.rs:159-168; the lattice lives in LDS, longer sentences are swept in segments
// cut at any position, the window of open end lists handed over; escape tiers behind it for a window wider than the tier)
// -> tokenize_global for whatever is left -> compact_tokens (tokens from per-sentence staging into sentence order; tok_tile_scan
// in front of it for huge batches).  The fused single-kernel design (fused.hip) is the fallback with a global-memory lattice and,
// with VBT_FUSED=1, an A/B reference.  Worker::tokenize is tokenize_serve: a resident wavefront that serves one Worker --
// generator + sweep of one sentence per doorbell, text and token records through pinned host memory.
//
// Bit-exactness notes: end lists are laid out with LDS atomics, so their order is arbitrary; every node
// carries its insertion sequence number and ties are broken towards the LARGEST sequence number, which is exactly what `<=` does
// in search_min_node (lattice.rs:141-146).  Candidates of positions the sweep never visits (unreachable or inside a skipped
// space run) are never inserted and never win as predecessors.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>

#include "engine.hpp"
#include "kernels.hpp"

#define HIP_CHECK(expr)                                                                               \
    do {                                                                                              \
        hipError_t e_ = (expr);                                                                       \
        if (e_ != hipSuccess)                                                                         \
            throw Error(VBT_ERR_DEVICE, std::string(#expr) + ": " + hipGetErrorString(e_));           \
    } while (0)

namespace vbt {
namespace {

template <typename T>
T* dev_upload(const std::vector<T>& v, std::vector<void*>& allocs) {
    T* p = nullptr;
    size_t bytes = std::max<size_t>(v.size() * sizeof(T), 16);
    HIP_CHECK(hipMalloc(reinterpret_cast<void**>(&p), bytes));
    allocs.push_back(p);
    if (!v.empty()) HIP_CHECK(hipMemcpy(p, v.data(), v.size() * sizeof(T), hipMemcpyHostToDevice));
    return p;
}

// hipMemset returns before the fill has run, and the fill is a null-stream operation: nothing orders it against kernels of the
// non-blocking streams every batch runs on.  Whatever is zeroed for a later launch is zeroed and waited for.
void zero_now(void* p, size_t bytes) {
    HIP_CHECK(hipMemsetAsync(p, 0, bytes, nullptr));
    HIP_CHECK(hipStreamSynchronize(nullptr));
}

uint32_t env_u32(const char* name, uint32_t dflt) {
    const char* s = std::getenv(name);
    return s && *s ? (uint32_t)std::strtoul(s, nullptr, 10) : dflt;
}

}  // namespace

// ------------------------------------------------------------------ Tokenizer

Tokenizer::Tokenizer(const Dictionary* dict, bool ignore_space, uint32_t max_grouping_len, int device) : dict_(dict) {
    setenv("GPU_MAX_HW_QUEUES", "10", 0);  // one stream per LDS tier (no effect once HIP is initialised)
    int count = 0;
    hipError_t e = hipGetDeviceCount(&count);
    if (e != hipSuccess || count == 0)
        throw Error(VBT_ERR_DEVICE, std::string("no HIP device available: ") + hipGetErrorString(e) +
                                        " (libvibrato_hip has no CPU fallback)");
    if (device < 0) HIP_CHECK(hipGetDevice(&device));
    if (device >= count) throw Error(VBT_ERR_INVALID_ARGUMENT, "device: index out of range");
    HIP_CHECK(hipSetDevice(device));
    device_ = device;
    {
        void* h = nullptr;
        HIP_CHECK(hipHostMalloc(&h, 64, hipHostMallocDefault));
        std::memset(h, 0, 64);
        density_host_ = static_cast<unsigned long long*>(h);
        void* d = nullptr;
        HIP_CHECK(hipHostGetDevicePointer(&d, h, 0));
        density_dev_ = static_cast<unsigned long long*>(d);
    }
    auto img0 = std::make_unique<DevImage>();
    DevDict& dev_ = img0->dev;  // image 0: the dictionary's own connection ids
    hipDeviceProp_t prop;
    HIP_CHECK(hipGetDeviceProperties(&prop, device));
    if (std::string(prop.gcnArchName).rfind("gfx950", 0) != 0)
        throw Error(VBT_ERR_DEVICE, std::string("device is ") + prop.gcnArchName + ", this library is built for gfx950 (MI355X) only");

    dev_.space_cateset = 0;
    if (ignore_space) {  // Tokenizer::ignore_space, tokenizer.rs:42-55
        int id = dict_->cate_id("SPACE");
        if (id < 0) throw Error(VBT_ERR_INVALID_ARGUMENT, "dict: SPACE is not defined in the input dictionary (i.e., char.def).");
        dev_.space_cateset = 1u << id;
    }
    dev_.max_grouping_len = max_grouping_len ? max_grouping_len : 0xFFFFFFFFu;  // tokenizer.rs:67-74
    try {
        upload_lexicon(dict_->system, dev_.sys);
        dev_.has_user = dict_->has_user ? 1 : 0;
        if (dict_->has_user) upload_lexicon(dict_->user, dev_.user);
        else dev_.user = dev_.sys;
        {   // lattice_lds addresses cells with 32-bit byte offsets through a buffer resource (checked before anything is allocated)
            const uint64_t bytes = (uint64_t)dict_->num_left * dict_->num_right * 2;
            if (bytes >= (1ull << 32)) throw Error(VBT_ERR_UNSUPPORTED, "connector: connection matrices of 4 GiB and more are not supported by the device image");
            dev_.matrix_bytes = (uint32_t)bytes;
        }
        if (dict_->conn_kind == kConnMatrix) {  // + 1 element: lattice_lds reads the aligned 32-bit word around a cell
            std::vector<int16_t> padded(dict_->matrix);
            padded.push_back(0);
            dev_.matrix = dev_upload(padded, allocs_);
        } else {
            // Compact connectors: evaluated once into a dense matrix (expand_connector).  i16 cells when every cost fits (the
            // released compact dictionaries: their costs are the matrix.def values of the full ones), else i32 cells -- the
            // reference's Raw / Dual cost is an i32 sum (raw_connector.rs:153-161, dual_connector.rs:267-279) -- read by the
            // kWide instances of the sweep kernels.
            const bool is_dual = dict_->conn_kind == kConnDual;
            const Scorer& sc = is_dual ? dict_->dual.scorer : dict_->raw.scorer;
            const size_t cells = (size_t)dict_->num_left * dict_->num_right;
            std::vector<void*> tmp;  // the compact structures are only needed by the expansion
            try {
                uint32_t* flag = nullptr;
                HIP_CHECK(hipMalloc(reinterpret_cast<void**>(&flag), 16));
                tmp.push_back(flag);
                zero_now(flag, 16);
                DevConnector c{};
                c.bases = dev_upload(sc.bases, tmp); c.checks = dev_upload(sc.checks, tmp); c.costs = dev_upload(sc.costs, tmp);
                c.n_bases = (uint32_t)sc.bases.size(); c.n_checks = (uint32_t)sc.checks.size();
                c.right_feats = dev_upload(is_dual ? dict_->dual.right_feats : dict_->raw.right_feats, tmp);
                c.left_feats = dev_upload(is_dual ? dict_->dual.left_feats : dict_->raw.left_feats, tmp);
                c.width = is_dual ? 8u : dict_->raw.width;
                if (is_dual) {
                    c.m = dev_upload(dict_->dual.matrix, tmp);
                    c.right_map = dev_upload(dict_->dual.right_map, tmp);
                    c.left_map = dev_upload(dict_->dual.left_map, tmp);
                    c.m_num_right = dict_->dual.m_num_right;
                }
                const dim3 grid((dict_->num_right + 255) / 256, dict_->num_left);
                int16_t* m = nullptr;
                HIP_CHECK(hipMalloc(reinterpret_cast<void**>(&m), (cells + 1) * 2));
                tmp.push_back(m);
                zero_now(m + cells, 2);
                kern::expand_connector_i16(grid, c, m, dict_->num_right, dict_->num_left, flag);
                uint32_t out_of_range = 0;
                HIP_CHECK(hipMemcpy(&out_of_range, flag, 4, hipMemcpyDeviceToHost));
                if (!out_of_range) {
                    tmp.pop_back();
                    allocs_.push_back(m);
                    dev_.matrix = m;
                } else {
                    if ((uint64_t)cells * 4 >= (1ull << 32))
                        throw Error(VBT_ERR_UNSUPPORTED, "connector: i32 connection matrices of 4 GiB and more are not supported by the device image");
                    (void)hipFree(m);
                    tmp.pop_back();
                    int32_t* w = nullptr;
                    HIP_CHECK(hipMalloc(reinterpret_cast<void**>(&w), (cells + 1) * 4));
                    tmp.push_back(w);
                    zero_now(w + cells, 4);
                    kern::expand_connector_i32(grid, c, w, dict_->num_right, dict_->num_left, flag);
                    HIP_CHECK(hipDeviceSynchronize());
                    tmp.pop_back();
                    allocs_.push_back(w);
                    dev_.matrix = reinterpret_cast<const int16_t*>(w);
                    dev_.matrix_wide = 1;
                    dev_.matrix_bytes = (uint32_t)(cells * 4);
                }
                for (void* p : tmp) (void)hipFree(p);
                tmp.clear();
            } catch (...) {
                for (void* p : tmp) (void)hipFree(p);
                throw;
            }
        }
        dev_.num_right = dict_->num_right;
        dev_.chr2inf = dev_upload(dict_->chr2inf, allocs_);
        {   // per code point: character class + trie codes in one 8-byte entry
            const size_t ms = dict_->system.mapper.size(), mu = dict_->has_user ? dict_->user.mapper.size() : 0;
            const size_t len = std::max<size_t>(65536, std::max(ms, mu));
            std::vector<uint2> cp(len + 1);
            for (size_t c = 0; c <= len; ++c) {
                const uint32_t info = dict_->chr2inf[c < 65536 && c < len ? c : 0];
                const uint32_t sc = c < len && c < ms ? dict_->system.mapper[c] : 0u, uc = c < len && c < mu ? dict_->user.mapper[c] : 0u;
                cp[c] = make_uint2(info, sc | (uc << 16));
            }
            dev_.cpinfo = dev_upload(cp, allocs_);
            dev_.cpinfo_len = (uint32_t)len;
        }
        dev_.unk_off = dev_upload(dict_->unk_offsets, allocs_);
        dev_.unk_entries = dev_upload(dict_->unk_entries, allocs_);
    } catch (...) {
        for (void* p : allocs_) (void)hipFree(p);
        throw;
    }
    // internal renumbering of the connection ids by measured usage (maybe_calibrate): on unless VBT_CONNID_REORDER=0
    if (env_u32("VBT_CONNID_REORDER", 1) == 0) calib_state_.store(3);
    calib_min_ = std::max<uint32_t>(1, env_u32("VBT_CONNID_MIN_SENTENCES", 2048));
    calib_sample_ = std::min<uint32_t>(1u << 20, std::max<uint32_t>(1, env_u32("VBT_CONNID_SAMPLE", 16384)));
    info_.min_sentences = calib_min_;
    if (calib_state_.load() == 0) {
        // the sample buffers: 512 bytes per sample sentence (the mean Japanese sentence is ~140), between 1 and 32 MiB
        try {
            s_cap_bytes_ = std::min<uint64_t>(32ull << 20, std::max<uint64_t>(1ull << 20, 512 * calib_sample_));
            HIP_CHECK(hipMalloc(reinterpret_cast<void**>(&s_text_), s_cap_bytes_ + 16)); allocs_.push_back(s_text_);
            HIP_CHECK(hipMalloc(reinterpret_cast<void**>(&s_offs_), (calib_sample_ + 1) * 8)); allocs_.push_back(s_offs_);
            HIP_CHECK(hipMalloc(reinterpret_cast<void**>(&s_src_), calib_sample_ * 4 + 16)); allocs_.push_back(s_src_);
            s_info_ = s_src_ + calib_sample_;
            HIP_CHECK(hipEventCreateWithFlags(reinterpret_cast<hipEvent_t*>(&calib_event_), hipEventDisableTiming));
            HIP_CHECK(hipStreamCreateWithFlags(reinterpret_cast<hipStream_t*>(&calib_stream_), hipStreamNonBlocking));
        } catch (...) {
            for (void* p : allocs_) (void)hipFree(p);
            if (calib_event_) (void)hipEventDestroy(reinterpret_cast<hipEvent_t>(calib_event_));
            throw;
        }
    }
    cur_.store(img0.get(), std::memory_order_release);
    images_.push_back(std::move(img0));
}

void Tokenizer::upload_lexicon(const Lexicon& lx, DevLexicon& out) {
    out.mapper = dev_upload(lx.mapper, allocs_);
    out.mapper_len = (uint32_t)lx.mapper.size();
    out.nodes = dev_upload(lx.nodes, allocs_);
    out.root_base = lx.nodes.empty() ? 0 : lx.nodes[0].base;
    out.entries = dev_upload(lx.entries, allocs_);
}

Tokenizer::~Tokenizer() {
    if (calib_thread_.joinable()) calib_thread_.join();  // (a calibration in flight reads the images and the sample buffers)
    (void)hipSetDevice(device_);
    if (calib_event_) (void)hipEventDestroy(reinterpret_cast<hipEvent_t>(calib_event_));
    if (calib_stream_) (void)hipStreamDestroy(reinterpret_cast<hipStream_t>(calib_stream_));
    for (auto& im : images_)
        for (void* p : im->allocs) (void)hipFree(p);
    for (void* p : allocs_) (void)hipFree(p);
    if (density_host_) (void)hipHostFree(density_host_);
}

double Tokenizer::candidates_per_byte() const {
    if (!density_host_) return 0.0;
    // (two separate 8-byte words written by one device thread, read without a lock: a torn pair is a slightly wrong ratio once)
    const unsigned long long c = __atomic_load_n(&density_host_[0], __ATOMIC_RELAXED), b = __atomic_load_n(&density_host_[1], __ATOMIC_RELAXED);
    return b ? (double)c / (double)b : 0.0;
}

const DevImage& Tokenizer::image_of(uint32_t epoch) const {
    std::lock_guard<std::mutex> g(img_mu_);
    for (const auto& im : images_)
        if (im->epoch == epoch) return *im;
    throw Error(VBT_ERR_INVALID_STATE, "no device image of that epoch");
}

ConnidReorderInfo Tokenizer::reorder_info() const {
    std::lock_guard<std::mutex> g(img_mu_);
    ConnidReorderInfo r = info_;
    r.state = (uint32_t)calib_state_.load(std::memory_order_acquire);
    r.epoch = image().epoch;
    return r;
}

// A copy of image 0 under a renumbering of the connection ids (perm_*[dictionary id] = device id; id 0 -- BOS / EOS -- stays 0):
// the matrix with rows and columns permuted on the device, the entry arrays with their id pairs translated on the host.
std::unique_ptr<DevImage> Tokenizer::renumbered_image(const std::vector<uint16_t>& pl, const std::vector<uint16_t>& pr, void* stream_) const {
    hipStream_t stream = reinterpret_cast<hipStream_t>(stream_);
    const DevImage& base = *images_[0];
    auto im = std::make_unique<DevImage>();
    im->dev = base.dev;
    im->perm_left = pl; im->perm_right = pr;
    try {
        auto entries = [&](const std::vector<Entry>& src) {
            std::vector<Entry> v(src);
            for (Entry& e : v) {
                const uint32_t l = e.left_right & 0xFFFFu, r = e.left_right >> 16;
                if (l >= pl.size() || r >= pr.size()) throw Error(VBT_ERR_INVALID_STATE, "entry with a connection id outside the matrix");
                e.left_right = (uint32_t)pl[l] | ((uint32_t)pr[r] << 16);
            }
            return dev_upload(v, im->allocs);
        };
        im->dev.sys.entries = entries(dict_->system.entries);
        im->dev.user.entries = dict_->has_user ? entries(dict_->user.entries) : im->dev.sys.entries;
        im->dev.unk_entries = entries(dict_->unk_entries);
        std::vector<uint16_t> il(pl.size()), ir(pr.size());  // device id -> dictionary id
        for (size_t i = 0; i < pl.size(); ++i) il[pl[i]] = (uint16_t)i;
        for (size_t i = 0; i < pr.size(); ++i) ir[pr[i]] = (uint16_t)i;
        std::vector<void*> tmp;
        try {
            const uint16_t* d_il = dev_upload(il, tmp);
            const uint16_t* d_ir = dev_upload(ir, tmp);
            const size_t cell = base.dev.matrix_wide ? 4 : 2;
            const size_t cells = (size_t)dict_->num_left * dict_->num_right;
            void* m = nullptr;
            HIP_CHECK(hipMalloc(&m, (cells + 1) * cell));
            im->allocs.push_back(m);
            zero_now(static_cast<char*>(m) + cells * cell, cell);
            kern::permute_matrix(stream, base.dev.matrix, m, base.dev.matrix_wide != 0, d_il, d_ir, dict_->num_left, dict_->num_right);
            HIP_CHECK(hipStreamSynchronize(stream));  // (not the device: a resident Worker kernel of another thread must not hold this up)
            im->dev.matrix = static_cast<const int16_t*>(m);
            for (void* p : tmp) (void)hipFree(p);
        } catch (...) {
            for (void* p : tmp) (void)hipFree(p);
            throw;
        }
    } catch (...) {
        for (void* p : im->allocs) (void)hipFree(p);
        throw;
    }
    return im;
}

// ------------------------------------------------------------------ Workspace

Workspace::Workspace(const Tokenizer& t, uint64_t max_s, uint64_t max_b) : tok(t), max_sentences(max_s), max_bytes(max_b) {
    try {
    HIP_CHECK(hipSetDevice(tok.device()));
    if (max_b >= 0xFFFFFFFFull || max_s >= 0xFFFFFFFFull) throw Error(VBT_ERR_INVALID_ARGUMENT, "workspace: batch too large (split it)");
    fused = env_u32("VBT_FUSED", 0) != 0;
    // LDS tiers (bytes per wave), ascending; the global-memory tier always follows
    {
        const char* e = std::getenv("VBT_TIERS");
        // (default: ONE 10 KiB tier that is also the segment tier -- 16 waves per CU, 4 per SIMD: lattice_lds is built for 128 VGPRs -- + two escape tiers)
        // (+ in front of it, where the build has the lean instance of the sweep, an 8 KiB tier for it -- five waves per SIMD -- for the whole
        // sentences that arrive with the generator's pass records; VBT_LEAN=0: without.  A second lean tier just under the segment tier's
        // size for the whole sentences of 8-10 KiB bought nothing: sweep 0.683 vs 0.682 ms, and every further tier costs the step ~0.015 ms)
        const bool lean_default = kern::lattice_has_lean() && env_u32("VBT_LEAN", 1) != 0;
        lean_tier_default = !(e && *e) && !fused && lean_default && env_u32("VBT_SEG_BYTES", kSegTierBytes) != 0;
        std::string spec = e && *e ? e : (fused ? "16384,32768,65536" : env_u32("VBT_SEG_BYTES", kSegTierBytes) ? (lean_default ? "8192,10240,49152,163840" : "10240,49152,163840") : "8192,12288,16384,24576,32768,49152,65536,163840");
        size_t pos = 0;
        while (pos < spec.size()) {
            size_t c = spec.find(',', pos);
            if (c == std::string::npos) c = spec.size();
            uint32_t v = (uint32_t)std::strtoul(spec.substr(pos, c - pos).c_str(), nullptr, 10);
            if (v < 256 || v > 163840 || (fused && v > 65536) || (!tiers.empty() && v <= tiers.back()) || tiers.size() >= kMaxTiers)
                throw Error(VBT_ERR_INVALID_ARGUMENT, "VBT_TIERS: expected up to 8 ascending LDS sizes in [256, 163840]");
            tiers.push_back(v);
            pos = c + 1;
        }
    }
    const size_t ns = std::max<uint64_t>(max_s, 1), nbts = std::max<uint64_t>(max_b, 1);
    auto alloc = [&](size_t bytes) {
        void* p = nullptr;
        HIP_CHECK(hipMalloc(&p, std::max<size_t>(bytes, 16)));
        pipe_allocs.push_back(p);
        return p;
    };
    d_tokens = static_cast<vbt_token_rec*>(alloc(nbts * sizeof(vbt_token_rec)));  // tokens <= chars <= bytes
    d_tok_stage = static_cast<vbt_token_rec*>(alloc((nbts + kSentenceSlack * ns + 1) * sizeof(vbt_token_rec)));  // per-sentence regions
    d_tile_sums = static_cast<uint32_t*>(alloc(((ns + kScanTile - 1) / kScanTile + 1) * 4));
    d_tok_off = static_cast<uint32_t*>(alloc(ns * 4));
    d_tok_cnt = static_cast<uint32_t*>(alloc(ns * 4));
    d_over = static_cast<uint32_t*>(alloc(2 * ns * 4 * (tiers.size() + 1 + kListsBehindTiers)));  // (+ 1: run() may put one more lean tier in front)  // two regions per list: the launch stream's and (VBT_EARLY_LONG=1) the long sentences' side streams'
    d_ctrl = static_cast<uint32_t*>(alloc((kCtrlWords + (size_t)kBlockCtrlWords) * 4));  // one block: cleared by one memset per batch
    d_cctrl = d_ctrl + kCtrlWords;
    if (const char* e = std::getenv("VBT_TIER_WAVES")) {  // experiment: fixed lattice grid per tier
        std::string spec = e;
        size_t pos = 0;
        while (pos < spec.size()) {
            size_t c = spec.find(',', pos);
            if (c == std::string::npos) c = spec.size();
            tier_waves.push_back((uint32_t)std::strtoul(spec.substr(pos, c - pos).c_str(), nullptr, 10));
            pos = c + 1;
        }
    }
    d_prof = static_cast<unsigned long long*>(alloc(2 * kProfSlots * kProfWords * 8));  // (second half: gen_long's phases in a VBT_GENLONG_PROF build)
    zero_now(d_prof, 2 * kProfSlots * kProfWords * 8);
    const uint64_t mb = env_u32("VBT_SCRATCH_MB", 0);
    // (fused fallback only: rare sentences; a one-sentence Worker must not pin 256 MiB)
    scratch_bytes = mb ? mb << 20 : std::max<uint64_t>(std::min<uint64_t>(256ull << 20, (16ull << 20) + 512 * nbts), 64 * nbts);
    d_scratch = static_cast<char*>(alloc(scratch_bytes));
    profile = env_u32("VBT_PROFILE", 0) != 0;
    for (auto& e : ev) HIP_CHECK(hipEventCreate(reinterpret_cast<hipEvent_t*>(&e)));
    // two-kernel pipeline buffers
    const size_t slots = nbts + kSentenceSlack * ns + 1;
    pipe.s_hdr = static_cast<uint4*>(alloc(ns * 16));
    if (!fused) {
        pipe.g_c2b = static_cast<uint16_t*>(alloc(slots * 2));
        pipe.g_pc = static_cast<uint4*>(alloc(slots * 16));
        pipe.node_factor = std::max<uint32_t>(1, env_u32("VBT_NODE_FACTOR", 8));  // candidate slots per input byte
        pipe.g_cand = static_cast<uint4*>(alloc((size_t)pipe.node_factor * slots * 16));
        pipe.g_hits = static_cast<uint4*>(alloc((size_t)pipe.node_factor * slots * 16));
        pipe.s_tier = static_cast<uint8_t*>(alloc(ns));
        // (the side streams of the LDS tiers are created when a launch sequence first needs one -- run(): only tier sets with several
        // tiers below the segment tier do.  Created here, every workspace took four streams, and with the HIP runtime's default of four
        // hardware queues the streams of the workspaces a pipelined host call alternates between all landed on ONE queue: their
        // kernels ran strictly one after the other, with a full drain at every switch.)
        streams.assign(tiers.size() + 1, nullptr);
        tier_events.assign(tiers.size() + 1, nullptr);
        HIP_CHECK(hipEventCreateWithFlags(reinterpret_cast<hipEvent_t*>(&ev_fork2), hipEventDisableTiming));

        kern::gen_set_max_lds(163840);
        if (tiers.back() > 65536)  // a single workgroup may use the CU's whole 160 KiB
        {
            kern::lattice_set_max_lds((int)tiers.back());
        }
    }
    } catch (...) {  // a failed hipMalloc / stream / event must not leak what was created before it
        release();
        throw;
    }
}

void Workspace::release() {
    for (void* p : pipe_allocs) (void)hipFree(p);
    pipe_allocs.clear();
    for (auto& e : ev) if (e) { (void)hipEventDestroy(reinterpret_cast<hipEvent_t>(e)); e = nullptr; }
    for (void* e : tier_events) if (e) (void)hipEventDestroy(reinterpret_cast<hipEvent_t>(e));
    tier_events.clear();
    if (ev_fork2) (void)hipEventDestroy(reinterpret_cast<hipEvent_t>(ev_fork2));
    ev_fork2 = nullptr;
    if (ev_fork_early) (void)hipEventDestroy(reinterpret_cast<hipEvent_t>(ev_fork_early));
    ev_fork_early = nullptr;
    for (void* st : streams) if (st) (void)hipStreamDestroy(reinterpret_cast<hipStream_t>(st));
    streams.clear();
}

Workspace::~Workspace() { release(); }

void Workspace::run(const uint8_t* d_text, const uint64_t* d_offsets, uint64_t n, uint64_t total_bytes, void* stream_, bool defer_pack) {
    if (n > max_sentences || total_bytes > max_bytes) throw Error(VBT_ERR_INVALID_ARGUMENT, "batch exceeds the workspace capacity");
    hipStream_t stream = reinterpret_cast<hipStream_t>(stream_);
    HIP_CHECK(hipSetDevice(tok.device()));
    // the tokenizer's first large batch has a sample of itself copied aside (two small kernels on this stream); a background thread
    // renumbers the connection ids of the device image by the usage it measures there and publishes the new image when it is ready
    if (!fused) tok.maybe_calibrate(d_text, d_offsets, n, total_bytes, stream_);
    const DevImage& image = tok.image();  // one image for every launch of this batch
    if (count_connids && image.epoch != count_epoch) {  // the counters on the device are in another image's ids: fold them first
        if (has_run) fold_connid_counts();  // (not "last_stream != nullptr": the null stream is a stream)
        count_epoch = image.epoch;
    }
    last_n = n;
    last_stream = stream_;
    has_run = true;
    HIP_CHECK(hipMemsetAsync(d_ctrl, 0, (kCtrlWords + (size_t)kBlockCtrlWords) * 4, stream));  // ctrl + cctrl
    if (packed_slot && n > packed_max_s) throw Error(VBT_ERR_INVALID_ARGUMENT, "batch has more sentences than the packed output slot holds");
    if (n == 0) {
        if (packed_slot) HIP_CHECK(hipMemsetAsync(packed_slot, 0, 32, stream));  // header of an empty result
        return;
    }
    // this batch's tiers: the workspace's, except that a batch of short sentences (at most 96 bytes on average: nearly every lattice fits)
    // gets a 6.5 KiB lean tier -- 24 of them on a CU, the lean instance's six waves per SIMD (80 VGPRs) -- instead of the 8 KiB one (20):
    // sentences of 5-20 characters 164 -> 170 M sentences/s; at the headline's 140 bytes per sentence the two are equal, and with both
    // tiers in the set the generator is slower by what the sweep gains (round 6, tools/dbg/tiers6_ab.sh).  VBT_TIERS set: as given.
    std::vector<uint32_t> tiers = this->tiers;
    uint32_t seg_bytes_default = kSegTierBytes;
    if (lean_tier_default && n && tiers.size() >= 2 && tiers[0] == 8192u && tiers[1] == kSegTierBytes) {
        // The default set {8 KiB lean, 10 KiB segments} is what DENSE lattices want (13 nodes per character: an 8 KiB segment holds too
        // few positions, dense law 31.3 -> 28.3 M sentences/s).  Everything else is swept faster with 8 KiB segments -- 20 instead of 16
        // of them on a CU, the slim instance's fifth wave per SIMD (96 VGPRs) -- behind a 7.5 KiB lean tier: headline 76.3 -> 79-80 M,
        // config 5 28.6 -> 29.2 M (tools/dbg/slim5_ab.sh, slim5_scan.sh).  Which it is the tokenizer knows from the batches before this
        // one (candidates_per_byte: 2.0 on running text, 4.4 on the dense law); the first batch is taken for running text.
        // VBT_TIERS / VBT_SEG_BYTES / VBT_TIER_ADAPT=0: the set as given.
        static const bool adapt = env_u32("VBT_TIER_ADAPT", 1) != 0;
        const bool dense = tok.candidates_per_byte() > 3.0;
        if (adapt && !dense) {
            // (two lean tiers, 6 and 7.5 KiB, where the sentences are long enough to fill both; their launches share one side stream:
            // headline 80.6-80.9 -> 81.6-82.4 M sentences/s, sweep 0.628-0.632 -> 0.607-0.612 ms; tools/dbg/tiers7_ab.sh, tiers8_ab.sh)
            // (batches of more than 256 bytes per sentence -- config 5: 330 -- are mostly long sentences in segments: one 7.5 KiB lean tier)
            const bool longs = total_bytes > 256 * n, shorts = total_bytes <= 96 * n;
            tiers[0] = longs ? 7680u : 6144u;
            tiers[1] = 8192u;
            if (!longs && !shorts && tiers.size() + 1 <= (size_t)kMaxTiers) tiers.insert(tiers.begin() + 1, 7680u);
            seg_bytes_default = 8192u;
        }
    }
    last_T = (uint32_t)tiers.size();
    const size_t T = tiers.size();
    const size_t stride = 2 * std::max<uint64_t>(max_sentences, 1);
    BatchArgs a = pipe;
    a.text = d_text; a.offsets = d_offsets; a.n = (uint32_t)n;
    a.tokens = d_tokens; a.tok_stage = d_tok_stage; a.tok_cap = (uint32_t)std::max<uint64_t>(max_bytes, 1);
    a.tok_off = d_tok_off; a.tok_cnt = d_tok_cnt; a.ctrl = d_ctrl; a.tile_sums = d_tile_sums;
    a.out_header = nullptr;
    if (packed_slot) {  // results straight into the caller's slot (set_packed_output)
        char* base = static_cast<char*>(packed_slot);
        a.out_header = reinterpret_cast<uint32_t*>(base);
        a.tok_off = reinterpret_cast<uint32_t*>(base + 32);
        a.tok_cnt = reinterpret_cast<uint32_t*>(base + 32 + 4 * packed_max_s);
        a.tokens = reinterpret_cast<vbt_token_rec*>(base + 32 + 8 * packed_max_s);
        a.tok_cap = (uint32_t)std::min<uint64_t>((packed_bytes - 32 - 8 * packed_max_s) / sizeof(vbt_token_rec), 0xFFFFFFFFull);
    }
    a.scratch = d_scratch; a.scratch_bytes = scratch_bytes;
    a.prof = profile ? d_prof : nullptr;
    a.lists = d_over; a.list_stride = (uint32_t)stride; a.n_tiers = (uint32_t)T;
    a.tier_prio = env_u32("VBT_TIER_PRIO", 3);
    {   // tier whose waves sweep longer sentences segment by segment (VBT_SEG_BYTES=0: off, sentences use the big tiers)
        const uint32_t seg_bytes = env_u32("VBT_SEG_BYTES", seg_bytes_default);
        a.seg_tier = 0xFFFFFFFFu;
        if (seg_bytes)
            for (size_t t = 0; t < T; ++t)
                if (tiers[t] >= seg_bytes) { a.seg_tier = (uint32_t)t; break; }
    }
    a.sid0 = 0; a.cctrl = d_cctrl; a.list_off = 0;
    last_seg_tier = a.seg_tier;
    // The tiers in front of the segment tier go to the lean instance of the sweep (whole sentences with the generator's records; everything
    // else -- segmented, counted, i32 cells -- is routed to the tiers behind them): VBT_LEAN=0 / VBT_LAT_PERSIST=1 keep the general
    // instance everywhere.  Without a segment tier only the first tier is lean.
    a.n_lean = 0;
    if (kern::lattice_has_lean() && env_u32("VBT_LEAN", 1) != 0 && !env_u32("VBT_LAT_PERSIST", 0) && !fused && T >= 2 && a.seg_tier != 0 && !count_connids && !image.dev.matrix_wide) {
        a.n_lean = a.seg_tier < T ? a.seg_tier : 1u;
        for (uint32_t t = 0; t < a.n_lean; ++t)
            if (tiers[t] > 65536u) { a.n_lean = t; break; }
    }
    a.density_out = fused || !lean_tier_default || env_u32("VBT_TIER_ADAPT", 1) == 0 ? nullptr : tok.density_slot_dev();  // (reported only where it is used)
    a.lid_count = count_connids ? d_connid : nullptr;
    a.rid_count = count_connids ? d_connid + tok.dict().num_left : nullptr;
    a.s_counted = count_connids ? d_counted : nullptr;
    if (count_connids) HIP_CHECK(hipMemsetAsync(d_counted, 0, n * 4, stream));
    for (size_t t = 0; t < T; ++t) a.tier_bytes[t] = tiers[t];
    const DevDict& D = image.dev;
    auto rec = [&](int i) { if (timing) HIP_CHECK(hipEventRecord(reinterpret_cast<hipEvent_t>(ev[i]), stream)); };
    auto over = [&](size_t t) { return d_over + t * stride; };
    auto waves_for = [&](uint32_t lds, uint64_t items) {
        const uint32_t per_cu = std::min<uint32_t>(32, std::max<uint32_t>(1, 163840 / lds));
        return (uint32_t)std::max<uint64_t>(1, std::min<uint64_t>(items, (uint64_t)per_cu * 256));
    };
    rec(0);
    {   // device-side input contract (offsets, UTF-8); a rejected batch is skipped by every kernel below
        const uint64_t work = std::max<uint64_t>(n, total_bytes / 8 + 1);
        kern::validate_batch((uint32_t)std::min<uint64_t>((work + 255) / 256, 1u << 16), stream, a, total_bytes);
    }
    if (fused) {
        auto count = [&](size_t t) { return d_cctrl + 2 * t; };
        auto cursor = [&](size_t t) { return d_cctrl + 2 * t + 1; };
        kern::fused_lds((uint32_t)n, tiers[0], stream, D, a, nullptr, nullptr, nullptr, over(0), count(0));
        rec(1);
        for (size_t t = 1; t < T; ++t)
            kern::fused_lds(waves_for(tiers[t], n), tiers[t], stream, D, a, over(t - 1), count(t - 1), cursor(t), over(t), count(t));
        kern::fused_global((uint32_t)std::min<uint64_t>(n, 1024), stream, D, a, over(T - 1), count(T - 1), cursor(T));
        rec(3);
    } else {
        // Stream plan.  The launch stream runs gen_candidates -> build_lists -> gen_candidates_large (gen_long: the sentences
        // that outgrew the bulk generator's LDS, one workgroup of several wavefronts each) and then forks one lattice_lds launch
        // per LDS tier onto the tier streams and joins them.
        // (the bulk generator keeps ~26 bytes of LDS per character: 5 KiB hold ~195 characters at 32 waves per CU)
        // (5 KiB: 32 waves x 5 KiB = the CU's 160 KiB; ~195 characters per sentence and room for ~130 more staged hits than 4 KiB -- config 5's
        // generator 1.46-1.51 -> 1.38-1.45 ms, the headline's 0.564 -> 0.557-0.562; 6 / 8 KiB cost occupancy: slower)
        const uint32_t gen_lds = env_u32("VBT_GEN_LDS", 5120);
        uint32_t gen_level_lds[kGenLevels] = {16384, 32768, 163840};  // the levels of gen_long (VBT_GEN_LEVELS=a,b,c): ~16 bytes per character
        if (const char* e = std::getenv("VBT_GEN_LEVELS")) {
            unsigned v[3];
            if (std::sscanf(e, "%u,%u,%u", &v[0], &v[1], &v[2]) == 3 && v[0] >= 4096 && v[0] < v[1] && v[1] < v[2] && v[2] <= 163840)
                for (int q = 0; q < 3; ++q) gen_level_lds[q] = v[q];
        }
        const uint32_t cn = (uint32_t)n, lb = (cn + 1023) / 1024;
        a.sid0 = 0; a.n = cn; a.cctrl = d_cctrl; a.list_off = 0; a.direct_push = 0; a.inline_lean = 0;
        for (int q = 0; q < kGenLevels; ++q) a.gen_level_bytes[q] = gen_level_lds[q] - 16;
        const uint32_t persist = env_u32("VBT_LAT_PERSIST", 0);
        auto launch_lattice = [&](const BatchArgs& a_, dim3 grid_, uint32_t lds_, hipStream_t st_, uint32_t tier_, uint32_t persistent_) {
            kern::lattice_lds(grid_.x, lds_, st_, D, a_, tier_, persistent_);
        };
        // (s_tier[] = 0xFF, "nothing routed yet", is written by validate_batch: one launch less per batch)
        // (VBT_GEN_SWEEP=1, experiment: the generator's wave sweeps its own sentence when it routed it to the lean tier)
        const bool gen_sweep = a.n_lean == 1 && env_u32("VBT_GEN_SWEEP", 0) != 0;
        a.inline_lean = gen_sweep ? 1u : 0u;
        last_inline = gen_sweep;
        if (gen_sweep) kern::gen_sweep(cn, tiers[0], stream, D, a);
        else kern::gen_candidates(cn, gen_lds, stream, D, a);
        kern::build_lists(lb, stream, a, -1);
        const size_t n_conc = a.seg_tier < T ? a.seg_tier + 1 : T;  // the tiers sentences are routed to up front: one launch each, side by side
        // The segment tier (the critical path: the longest sentences, then the escape tiers behind it) is launched on the launch
        // stream itself -- no event round trip before it starts nor before what follows it (VBT_MAIN_SEG=0: every tier on a side stream).
        const bool main_seg = env_u32("VBT_MAIN_SEG", 1) != 0 && a.seg_tier < T;
        // (every side stream costs the step a fork and a join across queues, ~0.02-0.03 ms: several lean tiers share ONE side stream, one
        // launch behind the other -- VBT_LEAN_ONE_STREAM=0: a stream each)
        static const bool lean_one = env_u32("VBT_LEAN_ONE_STREAM", 1) != 0;
        auto slot_of = [&](size_t t) { return lean_one && t < a.n_lean ? (size_t)0 : t; };
        // (the lean tiers on the launch stream and the segment tier on the side stream instead: sweep 0.604-0.614 -> 0.654-0.659 ms -- the long sentences' chain has to start first)
        auto is_main = [&](size_t t) { return main_seg && t == a.seg_tier; };
        auto launch_tier = [&](size_t t_, hipEvent_t after) {
            const size_t t = t_, si = slot_of(t_);
            const bool on_main = is_main(t);
            if (!on_main && !streams[si]) {
                hipStream_t new_stream;
                HIP_CHECK(hipStreamCreateWithFlags(&new_stream, hipStreamNonBlocking));
                streams[si] = new_stream;
                hipEvent_t new_event;
                HIP_CHECK(hipEventCreateWithFlags(&new_event, hipEventDisableTiming));
                tier_events[si] = new_event;
            }
            hipStream_t side = on_main ? stream : reinterpret_cast<hipStream_t>(streams[si]);
            if (!on_main) HIP_CHECK(hipStreamWaitEvent(side, after, 0));
            // one workgroup per list entry: the lists are built on the device, so the grid covers the whole batch and the
            // workgroups beyond a list's length exit at once (VBT_LAT_PERSIST=1: persistent waves with a work cursor)
            const uint32_t grid = !persist ? cn : (t < tier_waves.size() && tier_waves[t]) ? std::min<uint32_t>(tier_waves[t], waves_for(tiers[t], cn)) : waves_for(tiers[t], cn);
            // (where the lean instance runs -- the default build on i16 cells, no counting -- the segment tier gets the slim instance of the
            // general sweep: the same code without its C++ loop, `exact` mode and counting phase; VBT_SLIM=0: the general instance)
            static const bool slim_on = env_u32("VBT_SLIM", 1) != 0;
            if (t < a.n_lean) kern::lattice_lean(cn, tiers[t], side, D, a, (uint32_t)t);
            else if (slim_on && a.n_lean > 0 && !persist && tiers[t] <= 65536u) kern::lattice_slim(cn, tiers[t], side, D, a, (uint32_t)t);
            else launch_lattice(a, dim3(grid), tiers[t], side, (uint32_t)t, persist);
            if (t == a.seg_tier)
                for (size_t x = t + 1; x < T; ++x)
                    launch_lattice(a, dim3(waves_for(tiers[x], std::min<uint32_t>(cn, 4096))), tiers[x], side, (uint32_t)x, 1u);
            if (!on_main) HIP_CHECK(hipEventRecord(reinterpret_cast<hipEvent_t>(tier_events[si]), side));
        };
        // (VBT_LEAN_EARLY=1, a measured negative result kept as a knob: the lean tiers' lists are complete behind build_lists -- gen_long
        // files nothing there -- so their sweep could start here, next to the gen_long levels, in whose tail the machine idles for 55 us
        // of the headline step.  It does start, and its 100 k one-wave workgroups keep gen_long's 16-32 KiB workgroups off the CUs until
        // they have drained: headline 1.339 -> 1.348 ms, config 5 3.590 -> 3.575 ms.  Round 3 saw the same from the other side.)
        const bool lean_early = a.n_lean > 0 && env_u32("VBT_LEAN_EARLY", 0) != 0;
        if (lean_early) {
            if (!ev_fork_early) HIP_CHECK(hipEventCreateWithFlags(reinterpret_cast<hipEvent_t*>(&ev_fork_early), hipEventDisableTiming));
            HIP_CHECK(hipEventRecord(reinterpret_cast<hipEvent_t>(ev_fork_early), stream));
            for (size_t t = 0; t < a.n_lean && t < n_conc; ++t) launch_tier(t, reinterpret_cast<hipEvent_t>(ev_fork_early));
        }
        // gen_one files every sentence that outgrows it at the smallest level of gen_long that holds it; the levels run one after
        // the other on the launch stream: workgroups of 4 wavefronts (16 at the last level, which has a CU to itself), as many as
        // a CU's LDS and its 32 wave slots admit.  (Next to the bulk generator, on side streams, their 16-160 KiB workgroups do not
        // get onto a CU before the bulk launch has been dispatched: profiles/r03_long_first_experiment.md.)
        for (uint32_t lv = 1; lv <= (uint32_t)kGenLevels; ++lv) {
            const uint32_t lds = gen_level_lds[lv - 1];
            // (4 wavefronts per workgroup at the first level, 8 above it: 32 waves on a CU at either; config 5's generator 1.55 -> 1.45-1.48 ms against 4 everywhere)
            const uint32_t nw = lds > 65536 ? 16u : std::max<uint32_t>(1, std::min<uint32_t>(16, lv == 1 ? env_u32("VBT_GEN_WAVES1", 4) : env_u32("VBT_GEN_WAVES", 8)));
            const uint32_t per_cu = std::max<uint32_t>(1, std::min<uint32_t>(32 / nw, 163840 / lds));
            kern::gen_candidates_large(std::max<uint32_t>(1, std::min<uint32_t>(cn, per_cu * 256)), nw, lds, stream, D, a, lv);
        }
        rec(1);
        HIP_CHECK(hipEventRecord(reinterpret_cast<hipEvent_t>(ev_fork2), stream));
        // One launch per LDS tier up to the segment tier (the lean tiers in front of it may be running already).  The tiers above it
        // are escape tiers: nothing is routed to them up front, they take what the tier before them could not sweep (a window of end
        // lists wider than its LDS), so they are launched on the segment tier's stream, behind it.
        static const uint32_t skip_env = env_u32("VBT_SKIP_SWEEP", 0);
        const uint32_t skip_mode = count_connids ? 0u : skip_env;  // (the calibration's counting run stays whole: the probes time the renumbered image)  // (timing probes, results wrong: 1 = no sweep at all (tools/dbg/gen_ablate.py), 2 = the lean tiers only, 3 = all but the lean tiers)
        const bool skip_sweep = skip_mode == 1;
        // (largest tier first, on the shared lean stream too: the smaller lean tier's launch in front of the larger one's: sweep 0.612-0.622 -> 0.632-0.637 ms)
        for (size_t i = 0; i < n_conc; ++i) {
            const size_t t = n_conc - 1 - i;
            if (skip_sweep) break;
            if ((skip_mode == 2 && t >= a.n_lean) || (skip_mode == 3 && t < a.n_lean)) {  // (the join below still finds the tier's event recorded)
                if (!is_main(t) && tier_events[t]) HIP_CHECK(hipEventRecord(reinterpret_cast<hipEvent_t>(tier_events[t]), stream));
                continue;
            }
            if (lean_early && t < a.n_lean) continue;
            launch_tier(t, reinterpret_cast<hipEvent_t>(ev_fork2));
        }
        for (size_t t = 0; t < n_conc && !skip_sweep; ++t)
            if (!is_main(t) && slot_of(t) == t && tier_events[t]) HIP_CHECK(hipStreamWaitEvent(stream, reinterpret_cast<hipEvent_t>(tier_events[t]), 0));
        rec(3);
        // whatever the pipeline could not take: fused kernel, global-memory lattice (persistent waves with a work cursor; the list is
        // empty or a handful of sentences, and the kernel uses scratch memory: launching 1024 of them cost 15 us, 128 cost 6)
        kern::fused_global((uint32_t)std::min<uint64_t>(cn, env_u32("VBT_FB_WGS", 128)), stream, D, a, over(T), d_cctrl + 2 * T, d_cctrl + 2 * T + 1);
    }
    {   // pack the tokens in sentence order (tok_off, total)
        const uint32_t n_tiles = (uint32_t)((n + kScanTile - 1) / kScanTile);
        // (the totals per tile were added up by the kernels that emitted the tokens; their prefix is taken inside compact_tokens
        // unless the batch is huge or the caller packs later and wants the grand total first)
        const bool scan_kernel = defer_pack || n_tiles > 2048 || env_u32("VBT_PACK_SCAN", 0);
        if (scan_kernel) kern::tok_tile_scan(stream, a, d_tile_sums, n_tiles);
        if (!defer_pack) kern::compact_tokens(stream, a, d_tile_sums, n_tiles, scan_kernel ? 1u : 0u);
        last_args = a;
    }
    rec(2);
    HIP_CHECK(hipGetLastError());
}

void Workspace::serve(const uint8_t* h_text_dev, uint8_t* d_text, uint64_t* d_offsets, vbt_token_rec* tokens_out, uint32_t* ctl, uint32_t last_seq,
                      uint32_t idle_polls, void* stream_) {
    if (fused) throw Error(VBT_ERR_UNSUPPORTED, "the resident Worker kernel needs the two-kernel pipeline (unset VBT_FUSED)");
    hipStream_t stream = reinterpret_cast<hipStream_t>(stream_);
    last_n = 1;
    last_stream = stream_;
    BatchArgs a = pipe;
    a.text = d_text; a.offsets = d_offsets; a.n = 1;
    a.tokens = d_tokens; a.tok_stage = tokens_out; a.tok_cap = (uint32_t)std::max<uint64_t>(max_bytes, 1);
    a.tok_off = d_tok_off; a.tok_cnt = ctl + 1; a.ctrl = d_ctrl; a.tile_sums = d_tile_sums;  // (tile sums: written, never read on this path)
    a.scratch = d_scratch; a.scratch_bytes = scratch_bytes;
    a.prof = nullptr;
    a.lists = d_over; a.list_stride = (uint32_t)(2 * std::max<uint64_t>(max_sentences, 1)); a.n_tiers = 1;
    a.tier_prio = 0; a.seg_tier = 0; a.n_lean = 0; a.sid0 = 0; a.cctrl = d_cctrl; a.list_off = 0; a.direct_push = 0; a.inline_lean = 0;
    a.lid_count = nullptr; a.rid_count = nullptr; a.s_counted = nullptr; a.density_out = nullptr;
    constexpr uint32_t kOneLds = 65536;  // generator arrays (~26 B per character), then the lattice (whole up to ~400 characters, in segments beyond)
    a.tier_bytes[0] = kOneLds;
    const DevDict& D = tok.dev();  // (the image of the moment: a kernel that is started later may use a newer one)
    // (VBT_WORKER_MAX_SERVED: sentences after which the resident kernel leaves and is started again by the next call -- ~0.15 s of
    // residency at 35 us per call; 0 = unbounded)
    kern::tokenize_serve(kOneLds, stream, D, a, h_text_dev, ctl, last_seq, idle_polls, env_u32("VBT_WORKER_MAX_SERVED", 4096));
    HIP_CHECK(hipGetLastError());
}

void Workspace::set_packed_output(void* slot, uint64_t slot_bytes, uint64_t max_s) {
    if (slot) {
        if (fused) throw Error(VBT_ERR_UNSUPPORTED, "packed output needs the two-kernel pipeline (unset VBT_FUSED)");
        if ((reinterpret_cast<uintptr_t>(slot) & 7u) != 0) throw Error(VBT_ERR_INVALID_ARGUMENT, "packed output: the slot must be 8-byte aligned");
        if (slot_bytes < 32 + 8 * max_s) throw Error(VBT_ERR_INVALID_ARGUMENT, "packed output: the slot is smaller than its header and per-sentence arrays");
    }
    packed_slot = slot; packed_bytes = slot ? slot_bytes : 0; packed_max_s = slot ? max_s : 0;
}

void Workspace::pack_to(vbt_token_rec* out_tokens, uint32_t* out_off, uint32_t* out_cnt, void* stream_) {
    if (last_n == 0) return;
    const uint32_t n_tiles = (uint32_t)((last_n + kScanTile - 1) / kScanTile);
    const uint32_t wgs = std::max<uint32_t>(1, std::min<uint32_t>(n_tiles, env_u32("VBT_PACK_WGS", n_tiles)));
    kern::compact_tokens_out(wgs, reinterpret_cast<hipStream_t>(stream_), last_args, d_tile_sums, n_tiles, out_tokens, out_off, out_cnt);
    HIP_CHECK(hipGetLastError());
}

void Workspace::stats(vbt_call_stats* out) {
    HIP_CHECK(hipSetDevice(tok.device()));
    HIP_CHECK(hipStreamSynchronize(reinterpret_cast<hipStream_t>(last_stream)));
    uint32_t ctrl[kCtrlWords];
    std::vector<uint32_t> cc((size_t)kBlockCtrlWords);
    HIP_CHECK(hipMemcpy(ctrl, d_ctrl, sizeof(ctrl), hipMemcpyDeviceToHost));
    HIP_CHECK(hipMemcpy(cc.data(), d_cctrl, cc.size() * 4, hipMemcpyDeviceToHost));
    std::memset(out, 0, sizeof(*out));
    const size_t T = last_T ? last_T : tiers.size();  // (the tiers of the last run: its default set may have had one tier more)
    out->n_sentences = last_n;
    if (fused) {
        out->n_tier0 = last_n - cc[0];
        out->n_tier2 = cc[2 * (T - 1)];
        out->n_tier1 = last_n - out->n_tier0 - out->n_tier2;
    } else {
        // n_tier0: the tiers sentences are routed to up front (up to the segment tier; without one: the first tier), n_tier1: the tiers
        // behind them (escape tiers: what a sweep passed on), n_tier2: the global-memory kernel
        const size_t front = last_seg_tier < T ? last_seg_tier + 1 : 1;
        for (size_t t = 0; t < T; ++t) (t < front ? out->n_tier0 : out->n_tier1) += cc[2 * t];
        out->n_tier2 = cc[2 * T];
        if (last_inline) out->n_tier0 += cc[1];  // (gen_sweep: swept by the generator's own wave, counted by build_lists)
    }
    out->n_tokens = ctrl[kTotal];
    out->error_flags = ctrl[kError];
    if (std::getenv("VBT_DEBUG")) {
        std::fprintf(stderr, "[vbt] lattice fallbacks: arena=%u window=%u passes=%u no-cut=%u space-tail=%u interface/backtrace=%u; lists:", ctrl[26], ctrl[27], ctrl[29], ctrl[30], ctrl[31], ctrl[28]);
        for (size_t t = 0; t < T + kListsBehindTiers; ++t) std::fprintf(stderr, " %u", cc[2 * t]);
        std::fprintf(stderr, "; guard (VBT_GUARD builds): %u %u %u %u", ctrl[20], ctrl[21], ctrl[22], ctrl[23]);
        std::fprintf(stderr, "\n");
    }
    if (timing && last_n) {
        HIP_CHECK(hipEventElapsedTime(&out->ms_tier0, reinterpret_cast<hipEvent_t>(ev[0]), reinterpret_cast<hipEvent_t>(ev[1])));
        HIP_CHECK(hipEventElapsedTime(&out->ms_tier12, reinterpret_cast<hipEvent_t>(ev[1]), reinterpret_cast<hipEvent_t>(ev[3])));
        HIP_CHECK(hipEventElapsedTime(&out->ms_pack, reinterpret_cast<hipEvent_t>(ev[3]), reinterpret_cast<hipEvent_t>(ev[2])));
    }
}

void Workspace::enable_connid_counts(bool on) {
    HIP_CHECK(hipSetDevice(tok.device()));
    if (on && fused) throw Error(VBT_ERR_UNSUPPORTED, "connection-id counting needs the two-kernel pipeline (unset VBT_FUSED)");
    const size_t words = (size_t)tok.dict().num_left + tok.dict().num_right;
    if (on && !d_connid) {
        void* p = nullptr;
        HIP_CHECK(hipMalloc(&p, std::max<size_t>(words * 8, 16)));
        pipe_allocs.push_back(p);
        d_connid = static_cast<unsigned long long*>(p);
        zero_now(d_connid, words * 8);
        void* q = nullptr;
        HIP_CHECK(hipMalloc(&q, std::max<size_t>(max_sentences * 4, 16)));
        pipe_allocs.push_back(q);
        d_counted = static_cast<uint32_t*>(q);
    }
    count_connids = on;
}

// Device counters (device ids of image `count_epoch`) -> dictionary ids, added to out_*.
static void add_unmapped(const DevImage& im, const std::vector<uint64_t>& dl, const std::vector<uint64_t>& dr, uint64_t* lid, uint64_t* rid) {
    for (size_t i = 0; i < dl.size(); ++i) lid[i] += dl[im.perm_left.empty() ? i : im.perm_left[i]];
    for (size_t i = 0; i < dr.size(); ++i) rid[i] += dr[im.perm_right.empty() ? i : im.perm_right[i]];
}

void Workspace::fold_connid_counts() {
    if (!d_connid) return;
    HIP_CHECK(hipSetDevice(tok.device()));
    if (has_run) HIP_CHECK(hipStreamSynchronize(reinterpret_cast<hipStream_t>(last_stream)));
    const size_t nl = tok.dict().num_left, nr = tok.dict().num_right;
    std::vector<uint64_t> dl(nl), dr(nr);
    HIP_CHECK(hipMemcpy(dl.data(), d_connid, nl * 8, hipMemcpyDeviceToHost));
    HIP_CHECK(hipMemcpy(dr.data(), d_connid + nl, nr * 8, hipMemcpyDeviceToHost));
    zero_now(d_connid, (nl + nr) * 8);
    acc_lid.resize(nl, 0); acc_rid.resize(nr, 0);
    add_unmapped(tok.image_of(count_epoch), dl, dr, acc_lid.data(), acc_rid.data());
}

void Workspace::read_connid_counts(uint64_t* lid, uint64_t* rid, bool reset) {
    HIP_CHECK(hipSetDevice(tok.device()));
    if (!d_connid) throw Error(VBT_ERR_INVALID_STATE, "connection-id counting was never enabled");
    if (has_run) HIP_CHECK(hipStreamSynchronize(reinterpret_cast<hipStream_t>(last_stream)));
    const size_t nl = tok.dict().num_left, nr = tok.dict().num_right;
    std::vector<uint64_t> dl(nl), dr(nr);
    HIP_CHECK(hipMemcpy(dl.data(), d_connid, nl * 8, hipMemcpyDeviceToHost));
    HIP_CHECK(hipMemcpy(dr.data(), d_connid + nl, nr * 8, hipMemcpyDeviceToHost));
    // what the caller sees is in the dictionary's ids, whatever numbering the image that counted uses
    for (size_t i = 0; i < nl; ++i) lid[i] = i < acc_lid.size() ? acc_lid[i] : 0;
    for (size_t i = 0; i < nr; ++i) rid[i] = i < acc_rid.size() ? acc_rid[i] : 0;
    add_unmapped(tok.image_of(count_epoch), dl, dr, lid, rid);
    if (reset) {
        zero_now(d_connid, (nl + nr) * 8);
        acc_lid.clear(); acc_rid.clear();
    }
}

void Workspace::reset_connid_counts() {
    HIP_CHECK(hipSetDevice(tok.device()));
    if (!d_connid) return;
    if (has_run) HIP_CHECK(hipStreamSynchronize(reinterpret_cast<hipStream_t>(last_stream)));
    zero_now(d_connid, ((size_t)tok.dict().num_left + tok.dict().num_right) * 8);
    acc_lid.clear(); acc_rid.clear();
}

// ------------------------------------------------------------------ connection ids by usage

void Tokenizer::finish_calibration(bool done, bool give_up) const {
    {
        std::lock_guard<std::mutex> g(calib_mu_);
        // (a rejected sample -- error flags in its own run, an empty sample -- lets a later batch try again, three times at most)
        if (!done && !give_up && ++calib_attempts_ >= 3) give_up = true;
        calib_state_.store(done ? 2 : give_up ? 3 : 0, std::memory_order_release);
    }
    calib_cv_.notify_all();
}

bool Tokenizer::wait_calibration(int64_t timeout_ms) const {
    std::unique_lock<std::mutex> g(calib_mu_);
    auto idle = [&] { return calib_state_.load(std::memory_order_acquire) != 1; };
    if (timeout_ms < 0) { calib_cv_.wait(g, idle); return true; }
    return calib_cv_.wait_for(g, std::chrono::milliseconds(timeout_ms), idle);
}

void Tokenizer::maybe_calibrate(const uint8_t* d_text, const uint64_t* d_offsets, uint64_t n, uint64_t total_bytes, void* stream_) const {
    if (n < calib_min_ || calib_state_.load(std::memory_order_acquire) != 0) return;
    {
        std::lock_guard<std::mutex> g(calib_mu_);
        int idle = 0;
        if (!calib_state_.compare_exchange_strong(idle, 1)) return;  // another thread is at it: this batch runs on the current image
        if (calib_thread_.joinable()) calib_thread_.join();  // (an earlier attempt whose sample was rejected: it has left the state at 0)
    }
    // everything the caller's stream gets: two small kernels and an event record -- no allocation, no synchronisation, no host wait
    hipStream_t stream = reinterpret_cast<hipStream_t>(stream_);
    kern::sample_batch(stream, d_text, d_offsets, n, total_bytes, (uint32_t)calib_sample_, s_cap_bytes_, s_text_, s_offs_, s_src_, s_info_);
    if (hipGetLastError() != hipSuccess || hipEventRecord(reinterpret_cast<hipEvent_t>(calib_event_), stream) != hipSuccess) {
        finish_calibration(false, true);
        return;
    }
    try {
        calib_thread_ = std::thread([this] { background_calibration(); });
    } catch (const std::exception&) {
        finish_calibration(false, true);  // out of threads: the image stays as it is
    }
}

void Tokenizer::background_calibration() const {
    bool done = false, give_up = false;
    try {
        HIP_CHECK(hipSetDevice(device_));
        HIP_CHECK(hipEventSynchronize(reinterpret_cast<hipEvent_t>(calib_event_)));  // the sample is complete
        hipStream_t cs = reinterpret_cast<hipStream_t>(calib_stream_);
        uint32_t ns = 0;
        HIP_CHECK(hipMemcpyAsync(&ns, s_info_, 4, hipMemcpyDeviceToHost, cs));
        HIP_CHECK(hipStreamSynchronize(cs));
        uint64_t bytes = 0;
        if (ns) {
            HIP_CHECK(hipMemcpyAsync(&bytes, s_offs_ + ns, 8, hipMemcpyDeviceToHost, cs));
            HIP_CHECK(hipStreamSynchronize(cs));
        }
        done = ns != 0 && calibrate_sample(ns, bytes);
    } catch (const std::exception& e) {
        // an optimisation that cannot run (no memory for the sample's workspace, ...) must not fail anybody's batch: the image
        // stays as it is, for good
        if (std::getenv("VBT_DEBUG")) std::fprintf(stderr, "[vbt] connection-id renumbering given up: %s\n", e.what());
        (void)hipGetLastError();
        give_up = true;
    }
    finish_calibration(done, give_up);
}

void Tokenizer::calibrate_host(const uint8_t* text, const uint64_t* offsets, uint64_t n) const {
    for (;;) {
        const int st = calib_state_.load(std::memory_order_acquire);
        if (st == 2 || st == 3 || n == 0) return;
        if (st == 1) { wait_calibration(-1); continue; }
        std::lock_guard<std::mutex> g(calib_mu_);
        int idle = 0;
        if (calib_state_.compare_exchange_strong(idle, 1)) { if (calib_thread_.joinable()) calib_thread_.join(); break; }
    }
    bool done = false, give_up = false;
    try {
        HIP_CHECK(hipSetDevice(device_));
        // the sample, spread evenly over the caller's sentences (as sample_plan does it on the device), cut where the buffer ends
        const uint64_t want = std::min<uint64_t>(n, calib_sample_);
        std::vector<uint64_t> so(1, 0);
        std::vector<uint8_t> st;
        for (uint64_t j = 0; j < want; ++j) {
            const uint64_t src = (uint64_t)(((unsigned __int128)j * n) / want);
            if (offsets[src + 1] < offsets[src]) throw Error(VBT_ERR_INVALID_ARGUMENT, "offsets must be non-decreasing");
            const uint64_t len = offsets[src + 1] - offsets[src];
            if (so.back() + len > s_cap_bytes_) break;
            st.insert(st.end(), text + offsets[src], text + offsets[src] + len);
            so.push_back(so.back() + len);
        }
        const uint64_t ns = so.size() - 1;
        if (ns) {
            hipStream_t cs = reinterpret_cast<hipStream_t>(calib_stream_);
            if (!st.empty()) HIP_CHECK(hipMemcpyAsync(s_text_, st.data(), st.size(), hipMemcpyHostToDevice, cs));
            HIP_CHECK(hipMemcpyAsync(s_offs_, so.data(), so.size() * 8, hipMemcpyHostToDevice, cs));
            HIP_CHECK(hipStreamSynchronize(cs));
            done = calibrate_sample(ns, so.back());
        }
    } catch (const Error& e) {
        finish_calibration(false, e.code != VBT_ERR_INVALID_ARGUMENT);
        throw;
    } catch (...) {
        finish_calibration(false, true);
        throw;
    }
    finish_calibration(done, give_up);
}

// The counting sweep over the sample (s_text_ / s_offs_, ns sentences) on the calibration stream, the sort, the renumbered image.
bool Tokenizer::calibrate_sample(uint64_t ns, uint64_t bytes) const {
    const auto t0 = std::chrono::steady_clock::now();
    const size_t nl = dict_->num_left, nr = dict_->num_right;
    std::vector<uint64_t> lid(nl), rid(nr);
    {
        Workspace ws(*this, ns, bytes);  // (its run() comes back to maybe_calibrate and finds the state "running")
        ws.enable_connid_counts(true);
        ws.run(s_text_, s_offs_, ns, bytes, calib_stream_);
        vbt_call_stats st;
        ws.stats(&st);
        if (st.error_flags) return false;
        ws.read_connid_counts(lid.data(), rid.data(), false);
        if (std::getenv("VBT_DEBUG")) {
            uint64_t sl = 0, sr = 0;
            for (uint64_t v : lid) sl += v;
            for (uint64_t v : rid) sr += v;
            std::fprintf(stderr, "[vbt] calibration sample: %llu sentences, %llu bytes, %llu tokens, counted pairs %llu / %llu\n", (unsigned long long)ns,
                         (unsigned long long)bytes, (unsigned long long)st.n_tokens, (unsigned long long)sl, (unsigned long long)sr);
        }
    }
    // ids by descending count, then ascending id (ConnIdCounter::compute_probs, mapper.rs:108-146); id 0 (BOS / EOS) stays
    auto order = [](const std::vector<uint64_t>& cnt, std::vector<uint16_t>& perm) {
        std::vector<uint32_t> ids(cnt.size() > 0 ? cnt.size() - 1 : 0);
        for (size_t i = 0; i < ids.size(); ++i) ids[i] = (uint32_t)i + 1;
        std::stable_sort(ids.begin(), ids.end(), [&](uint32_t a, uint32_t b) { return cnt[a] > cnt[b]; });
        perm.assign(cnt.size(), 0);
        uint32_t moved = 0;
        for (size_t k = 0; k < ids.size(); ++k) { perm[ids[k]] = (uint16_t)(k + 1); moved += ids[k] != k + 1; }
        return moved;
    };
    std::vector<uint16_t> pl, pr;
    const uint32_t ml = order(lid, pl), mr = order(rid, pr);
    std::unique_ptr<DevImage> im;
    if (ml || mr) {
        im = renumbered_image(pl, pr, calib_stream_);
        im->epoch = 1;
    }
    std::lock_guard<std::mutex> g(img_mu_);
    info_.sample_sentences = ns;
    info_.moved_left = ml; info_.moved_right = mr;
    if (im) {
        const DevImage* p = im.get();
        images_.push_back(std::move(im));
        cur_.store(p, std::memory_order_release);
    }
    info_.ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
    return true;
}

void Workspace::read_profile(uint64_t* out, bool reset) {
    HIP_CHECK(hipSetDevice(tok.device()));
    HIP_CHECK(hipStreamSynchronize(reinterpret_cast<hipStream_t>(last_stream)));
    std::vector<uint64_t> h((size_t)kProfSlots * kProfWords);
    unsigned long long* const half = d_prof + (env_u32("VBT_PROF_HALF", 0) ? h.size() : 0);
    HIP_CHECK(hipMemcpy(h.data(), half, h.size() * 8, hipMemcpyDeviceToHost));
    for (int i = 0; i < kProfWords; ++i) out[i] = 0;
    for (int k = 0; k < kProfSlots; ++k)
        for (int i = 0; i < kProfWords; ++i) out[i] += h[(size_t)k * kProfWords + i];
    if (reset) zero_now(half, h.size() * 8);
}

}  // namespace vbt

