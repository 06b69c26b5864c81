// Command-line counterpart of the reference's C++ example (examples/cpp/gptneox/gptneox_example.cc:36-470): a second
// caller of the engine without Python.  Reads an ini file with the reference's sections, loads a converted checkpoint
// (`.bin` / `.q.bin` + `.s.bin` files as huggingface_convert.py / quant_and_save.py -- here convert.py -- write them, C++
// loader of the reference: GptNeoXDecoderLayerWeight.cc:169-260, GptNeoXWeight.cc), runs the prompts of a csv file
// through include/ftcf.h and writes the token ids to `out`, one row per request.
//
//   gptneox_example <config.ini> [--start_ids <file.csv>] [--out <file>]
//                   [--rendezvous <dir>] [--exchange rccl|host] [--rank r --world n] [--device d]
//
// TENSOR PARALLEL (examples/cpp/gptneox/gptneox_example.cc:399-411 runs under mpirun, one rank per GPU): start
// `tensor_para_size` copies of this program, e.g. `torchrun --nproc-per-node N ... gptneox_example cfg.ini --rendezvous /tmp/r`
// or `mpirun -n N` -- rank and world size come from RANK / WORLD_SIZE (torchrun), OMPI_COMM_WORLD_RANK / _SIZE (mpirun) or the
// flags; the device from LOCAL_RANK, else rank % devices.  The ranks meet in a DIRECTORY (no MPI, no torch in this program):
// every exchange between them is an all-gather of small files there -- the RCCL unique id (nccl_utils.cc ftNcclInitialize's
// MPI_Bcast), or, with `--exchange host` / FTCF_TP_EXCHANGE=host, everything a host-exchange communicator needs
// (include/ftcf.h ftcf_comm_init_host_exchange: lets the ranks share one device -- how tests/test_gpu_cli.py runs TP = 2 on a
// one-GPU box).  Rank r loads the converter's `.r.bin` shards; rank 0 writes `out`.
//
// ini (same keys as examples/cpp/gptneox/gptneox_config.ini):
//   [ft_instance_hyperparameter]  model_name, model_dir, tensor_para_size (1 here: one process, one GPU), int8_mode
//   [request]                     request_batch_size, request_output_len, beam_width (1), beam_search_diversity_rate, len_penalty, top_k, top_p, temperature,
//                                 repetition_penalty
//   [<model_name>]                head_num, size_per_head, inter_size, vocab_size, decoder_layers, rotary_embedding,
//                                 start_id, end_id, use_gptj_residual, weight_data_type (fp32 | fp16)
//   (the model section may instead live in <model_dir>/config.ini as [gptneox] with num_layer, the converter's output)
#include <hip/hip_runtime.h>
#include <sys/stat.h>
#include <unistd.h>

#include <chrono>
#include <thread>

#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <map>
#include <sstream>
#include <stdexcept>
#include <string>
#include <vector>

#include "ftcf.h"

namespace {

using Ini = std::map<std::string, std::map<std::string, std::string>>;

std::string trim(const std::string& s)
{
    const size_t a = s.find_first_not_of(" \t\r\n");
    if (a == std::string::npos) {
        return "";
    }
    const size_t b = s.find_last_not_of(" \t\r\n");
    return s.substr(a, b - a + 1);
}

// ---- the ranks' meeting place: an all-gather of small files in a shared directory ---------------------------------
struct Rendezvous {
    std::string dir;
    int         rank = 0, world = 1;
    long        seq = 0;
    double      timeout_s = 600.0;

    std::string name(long k, int r) const { return dir + "/ag." + std::to_string(k) + "." + std::to_string(r); }
    // recv[r * bytes ..] = rank r's send; 0 on success (the signature of ftcf_host_allgather_fn)
    static int allgather(void* user, const void* send, void* recv, size_t bytes)
    {
        auto* self = static_cast<Rendezvous*>(user);
        try {
            const long k = self->seq++;
            if (k >= 2) {  // (everybody has read call k - 2: a rank enters call k only after reading all of k - 1)
                (void)unlink(self->name(k - 2, self->rank).c_str());
            }
            const std::string mine = self->name(k, self->rank), tmp = mine + ".tmp";
            {
                std::ofstream f(tmp, std::ios::binary);
                f.write(static_cast<const char*>(send), (std::streamsize)bytes);
                if (!f.good()) {
                    throw std::runtime_error("cannot write " + tmp);
                }
            }
            if (rename(tmp.c_str(), mine.c_str()) != 0) {
                throw std::runtime_error("cannot publish " + mine);
            }
            const auto t0 = std::chrono::steady_clock::now();
            for (int r = 0; r < self->world; r++) {
                const std::string fn = self->name(k, r);
                for (;;) {
                    struct stat st;
                    if (stat(fn.c_str(), &st) == 0 && (size_t)st.st_size == bytes) {
                        std::ifstream f(fn, std::ios::binary);
                        f.read(static_cast<char*>(recv) + (size_t)r * bytes, (std::streamsize)bytes);
                        if (f.good()) {
                            break;
                        }
                    }
                    if (std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() > self->timeout_s) {
                        throw std::runtime_error("rank " + std::to_string(r) + " did not arrive at " + fn);
                    }
                    std::this_thread::sleep_for(std::chrono::microseconds(200));
                }
            }
            return 0;
        }
        catch (const std::exception& e) {
            fprintf(stderr, "[ERROR] rendezvous: %s\n", e.what());
            return 1;
        }
    }
};

int env_int(const char* a, const char* b, int dflt)
{
    if (const char* v = getenv(a)) {
        return atoi(v);
    }
    if (b) {
        if (const char* v = getenv(b)) {
            return atoi(v);
        }
    }
    return dflt;
}

Ini read_ini(const std::string& path)
{
    std::ifstream f(path);
    if (!f.is_open()) {
        throw std::runtime_error("cannot open " + path);
    }
    Ini         ini;
    std::string line, sec;
    while (std::getline(f, line)) {
        const size_t c = line.find_first_of("#;");
        if (c != std::string::npos) {
            line = line.substr(0, c);
        }
        line = trim(line);
        if (line.empty()) {
            continue;
        }
        if (line.front() == '[' && line.back() == ']') {
            sec = trim(line.substr(1, line.size() - 2));
            continue;
        }
        const size_t e = line.find('=');
        if (e != std::string::npos) {
            ini[sec][trim(line.substr(0, e))] = trim(line.substr(e + 1));
        }
    }
    return ini;
}

std::string get(const Ini& ini, const std::string& sec, const std::string& key, const char* dflt = nullptr)
{
    auto s = ini.find(sec);
    if (s != ini.end()) {
        auto k = s->second.find(key);
        if (k != s->second.end()) {
            return k->second;
        }
    }
    if (!dflt) {
        throw std::runtime_error("missing [" + sec + "] " + key);
    }
    return dflt;
}

void hip_check(hipError_t e, const char* what)
{
    if (e != hipSuccess) {
        throw std::runtime_error(std::string(what) + ": " + hipGetErrorString(e));
    }
}
void ftcf_check(int rc, const char* what)
{
    if (rc != 0) {
        throw std::runtime_error(std::string(what) + ": " + ftcf_last_error());
    }
}

// fp32 -> fp16 bits, round to nearest even (the checkpoint's fp32 files are cast to the inference type, fp16)
uint16_t f32_to_f16(float f)
{
    uint32_t x;
    memcpy(&x, &f, 4);
    const uint32_t sign = (x >> 16) & 0x8000u;
    const int32_t  exp  = (int32_t)((x >> 23) & 0xff) - 127 + 15;
    uint32_t       man  = x & 0x7fffffu;
    if (((x >> 23) & 0xff) == 0xff) {
        return (uint16_t)(sign | 0x7c00u | (man ? 0x200u : 0));
    }
    if (exp >= 31) {
        return (uint16_t)(sign | 0x7c00u);
    }
    if (exp <= 0) {
        if (exp < -10) {
            return (uint16_t)sign;
        }
        man |= 0x800000u;
        const int      shift = 14 - exp;
        uint32_t       h     = man >> shift;
        const uint32_t rem   = man & ((1u << shift) - 1), half = 1u << (shift - 1);
        if (rem > half || (rem == half && (h & 1))) {
            h++;
        }
        return (uint16_t)(sign | h);
    }
    uint32_t       h   = ((uint32_t)exp << 10) | (man >> 13);
    const uint32_t rem = man & 0x1fffu;
    if (rem > 0x1000u || (rem == 0x1000u && (h & 1))) {
        h++;
    }
    return (uint16_t)(sign | h);
}

std::vector<char> read_file(const std::string& path)
{
    std::ifstream f(path, std::ios::binary | std::ios::ate);
    if (!f.is_open()) {
        throw std::runtime_error("cannot open " + path);
    }
    const std::streamsize n = f.tellg();
    f.seekg(0);
    std::vector<char> buf((size_t)n);
    f.read(buf.data(), n);
    return buf;
}

struct DeviceBlobs {
    std::vector<void*> ptrs;
    ~DeviceBlobs()
    {
        for (void* p : ptrs) {
            (void)hipFree(p);
        }
    }
    void* upload(const void* host, size_t bytes)
    {
        void* d = nullptr;
        hip_check(hipMalloc(&d, bytes ? bytes : 16), "hipMalloc");
        if (bytes && host) {  // host == NULL: allocate only
            hip_check(hipMemcpy(d, host, bytes, hipMemcpyHostToDevice), "hipMemcpy");
        }
        ptrs.push_back(d);
        return d;
    }
};

// a weight file of `count` elements in the checkpoint dtype -> fp16 device tensor
void* load_fp16(DeviceBlobs& dev, const std::string& path, size_t count, bool file_is_fp16)
{
    const std::vector<char> raw = read_file(path);
    const size_t            esz = file_is_fp16 ? 2 : 4;
    if (raw.size() != count * esz) {
        throw std::runtime_error(path + ": expected " + std::to_string(count) + " elements, file holds "
                                 + std::to_string(raw.size() / esz));
    }
    if (file_is_fp16) {
        return dev.upload(raw.data(), raw.size());
    }
    std::vector<uint16_t> h(count);
    const float*          src = reinterpret_cast<const float*>(raw.data());
    for (size_t i = 0; i < count; i++) {
        h[i] = f32_to_f16(src[i]);
    }
    return dev.upload(h.data(), count * 2);
}

}  // namespace

int main(int argc, char** argv)
{
    try {
        if (argc < 2) {
            fprintf(stderr, "usage: %s <config.ini> [--start_ids <file.csv>] [--out <file>] [--rendezvous <dir>] [--exchange rccl|host] "
                            "[--rank r --world n] [--device d]\n", argv[0]);
            return 2;
        }
        std::string ini_path = argv[1], ids_path = "start_ids.csv", out_path = "out";
        std::string rdv_dir = getenv("FTCF_RENDEZVOUS") ? getenv("FTCF_RENDEZVOUS") : "";
        std::string exchange = (getenv("FTCF_TP_EXCHANGE") && !strcmp(getenv("FTCF_TP_EXCHANGE"), "host")) ? "host" : "rccl";
        int rank = env_int("RANK", "OMPI_COMM_WORLD_RANK", 0), world = env_int("WORLD_SIZE", "OMPI_COMM_WORLD_SIZE", 1);
        int device_flag = -1;
        for (int i = 2; i + 1 < argc; i += 2) {
            if (!strcmp(argv[i], "--start_ids")) {
                ids_path = argv[i + 1];
            }
            else if (!strcmp(argv[i], "--out")) {
                out_path = argv[i + 1];
            }
            else if (!strcmp(argv[i], "--rendezvous")) {
                rdv_dir = argv[i + 1];
            }
            else if (!strcmp(argv[i], "--exchange")) {
                exchange = argv[i + 1];
            }
            else if (!strcmp(argv[i], "--rank")) {
                rank = atoi(argv[i + 1]);
            }
            else if (!strcmp(argv[i], "--world")) {
                world = atoi(argv[i + 1]);
            }
            else if (!strcmp(argv[i], "--device")) {
                device_flag = atoi(argv[i + 1]);
            }
        }
        const Ini         ini   = read_ini(ini_path);
        const std::string inst  = "ft_instance_hyperparameter";
        const std::string name  = get(ini, inst, "model_name");
        const std::string mdir  = get(ini, inst, "model_dir");
        const int         tp    = std::stoi(get(ini, inst, "tensor_para_size", "1"));
        const int         int8  = std::stoi(get(ini, inst, "int8_mode", "0"));
        if (std::stoi(get(ini, inst, "pipeline_para_size", "1")) != 1) {
            throw std::runtime_error("pipeline_para_size must be 1");
        }
        if (tp != world || rank < 0 || rank >= world) {
            throw std::runtime_error("tensor_para_size = " + std::to_string(tp) + " needs that many ranks (this is rank "
                                     + std::to_string(rank) + " of " + std::to_string(world) + ": torchrun / mpirun / --rank --world)");
        }
        if (tp > 1 && rdv_dir.empty()) {
            throw std::runtime_error("tensor_para_size > 1 needs --rendezvous <dir> (or FTCF_RENDEZVOUS): where the ranks meet");
        }
        // model hyper-parameters: [<model_name>] of the main ini, else [gptneox] of <model_dir>/config.ini
        Ini         mini = ini;
        std::string msec = name;
        if (!ini.count(name)) {
            mini = read_ini(mdir + "/config.ini");
            msec = "gptneox";
        }
        const int  nh = std::stoi(get(mini, msec, "head_num")), dh = std::stoi(get(mini, msec, "size_per_head"));
        const int  V = std::stoi(get(mini, msec, "vocab_size"));
        const int  L = std::stoi(mini[msec].count("decoder_layers") ? get(mini, msec, "decoder_layers") : get(mini, msec, "num_layer"));
        const int  rot = std::stoi(get(mini, msec, "rotary_embedding"));
        const int  start_id = std::stoi(get(mini, msec, "start_id", "0")), end_id = std::stoi(get(mini, msec, "end_id"));
        const int  H = nh * dh;
        const int  I = std::stoi(get(mini, msec, "inter_size", std::to_string(4 * H).c_str()));
        const bool gptj = std::stoi(get(mini, msec, "use_gptj_residual", "1")) != 0;
        const std::string wdt = get(mini, msec, "weight_data_type", "fp32");
        const bool        f16 = (wdt == "fp16" || wdt == "float16");

        const int   B       = std::stoi(get(ini, "request", "request_batch_size"));
        const int   out_len = std::stoi(get(ini, "request", "request_output_len"));
        const int   beam    = std::stoi(get(ini, "request", "beam_width", "1"));
        const int   top_k   = std::stoi(get(ini, "request", "top_k", "1"));
        const float top_p = std::stof(get(ini, "request", "top_p", "0")), temp = std::stof(get(ini, "request", "temperature", "1"));
        const float rep     = std::stof(get(ini, "request", "repetition_penalty", "1"));
        const float div_rate = std::stof(get(ini, "request", "beam_search_diversity_rate", "0"));
        const float len_pen  = std::stof(get(ini, "request", "len_penalty", "0"));

        // ---- prompts (gpt_example_utils.cc:27-100: one csv row per request, short rows padded with end_id, missing rows
        //      replaced by end_id rows) ----
        std::vector<std::vector<int>> rows;
        {
            std::ifstream f(ids_path);
            if (!f.is_open()) {
                throw std::runtime_error("cannot open " + ids_path);
            }
            std::string line;
            while (std::getline(f, line)) {
                std::stringstream ls(line);
                std::string       v;
                std::vector<int>  r;
                while (std::getline(ls, v, ',')) {
                    if (!trim(v).empty()) {
                        r.push_back(std::stoi(v));
                    }
                }
                if (!r.empty()) {
                    rows.push_back(r);
                }
            }
        }
        if (rows.empty()) {
            throw std::runtime_error("no prompts in " + ids_path);
        }
        size_t S = 0;
        for (auto& r : rows) {
            S = std::max(S, r.size());
        }
        std::vector<int> ids((size_t)B * S, end_id), lens(B, (int)S);
        for (int b = 0; b < B; b++) {
            if (b < (int)rows.size()) {
                std::copy(rows[b].begin(), rows[b].end(), ids.begin() + (size_t)b * S);
                lens[b] = (int)rows[b].size();
            }
        }

        int ndev = 0;
        hip_check(hipGetDeviceCount(&ndev), "hipGetDeviceCount");
        const int device = device_flag >= 0 ? device_flag : env_int("LOCAL_RANK", "OMPI_COMM_WORLD_LOCAL_RANK", rank) % (ndev > 0 ? ndev : 1);
        hip_check(hipSetDevice(device), "hipSetDevice");
        const int  hl = H / tp, il = I / tp;  // this rank's shard (huggingface_convert.py:35-81: QKV / FFN1 columns, out-proj / FFN2 rows)
        const std::string rk = "." + std::to_string(rank);
        // ---- weights in the order of GptNeoXOp.h:121-174 ----
        DeviceBlobs              dev;
        std::vector<const void*> w((size_t)12 * L + 4, nullptr), q, sc;
        static const char* names[12] = {"input_layernorm.bias", "input_layernorm.weight", "attention.query_key_value.weight.0",
                                        "attention.query_key_value.bias.0", "attention.dense.weight.0", "attention.dense.bias",
                                        "mlp.dense_h_to_4h.weight.0", "mlp.dense_h_to_4h.bias.0", "mlp.dense_4h_to_h.weight.0",
                                        "mlp.dense_4h_to_h.bias", "post_attention_layernorm.bias",
                                        "post_attention_layernorm.weight"};
        const size_t count[12] = {(size_t)H, (size_t)H, (size_t)H * 3 * hl, (size_t)3 * hl, (size_t)hl * H, (size_t)H,
                                  (size_t)H * il, (size_t)il, (size_t)il * H, (size_t)H, (size_t)H, (size_t)H};
        const size_t kdim[12]  = {0, 0, (size_t)H, 0, (size_t)hl, 0, (size_t)H, 0, (size_t)il, 0, 0, 0};
        const size_t ndim[12]  = {0, 0, (size_t)3 * hl, 0, (size_t)H, 0, (size_t)il, 0, (size_t)H, 0, 0, 0};
        if (int8) {
            q.assign((size_t)4 * L, nullptr);
            sc.assign((size_t)4 * L, nullptr);
        }
        for (int l = 0; l < L; l++) {
            const std::string base = mdir + "/model.layers." + std::to_string(l) + ".";
            for (int g = 0; g < 12; g++) {
                if (g == 5 && gptj) {
                    continue;  // empty slot (GptNeoXOp.h:137)
                }
                std::string fn = names[g];
                if (fn.size() > 2 && fn.compare(fn.size() - 2, 2, ".0") == 0) {
                    fn = fn.substr(0, fn.size() - 2) + rk;  // the converter's per-rank files
                }
                if (g == 9 && gptj) {
                    fn = "mlp.attention.bias.sum";  // attn-out bias + ffn2 bias (already divided by TP), written by the converter
                }
                const bool kernel = kdim[g] != 0;
                if (kernel && int8) {
                    const int               j   = g / 2 - 1;  // 2,4,6,8 -> 0..3
                    const std::vector<char> raw = read_file(base + fn + ".q.bin");
                    if (raw.size() != count[g]) {
                        throw std::runtime_error(base + fn + ".q.bin: wrong size");
                    }
                    q[(size_t)j * L + l]  = dev.upload(raw.data(), raw.size());
                    sc[(size_t)j * L + l] = load_fp16(dev, base + fn + ".s.bin", ndim[g], f16);
                    continue;
                }
                w[(size_t)g * L + l] = load_fp16(dev, base + fn + ".bin", count[g], f16);
            }
        }
        w[(size_t)12 * L]     = load_fp16(dev, mdir + "/model.wte.bin", (size_t)V * H, f16);
        w[(size_t)12 * L + 1] = load_fp16(dev, mdir + "/model.final_layernorm.weight.bin", H, f16);
        w[(size_t)12 * L + 2] = load_fp16(dev, mdir + "/model.final_layernorm.bias.bin", H, f16);
        w[(size_t)12 * L + 3] = load_fp16(dev, mdir + "/model.lm_head.weight.bin", (size_t)V * H, f16);

        hipStream_t stream = nullptr;
        hip_check(hipStreamCreateWithFlags(&stream, hipStreamNonBlocking), "hipStreamCreate");
        ftcf_gptneox_config cfg{};
        cfg.head_num = nh;
        cfg.size_per_head = dh;
        cfg.inter_size = I;
        cfg.num_layer = L;
        cfg.vocab_size = V;
        cfg.rotary_embedding_dim = rot;
        cfg.start_id = start_id;
        cfg.end_id = end_id;
        // ---- tensor-parallel communicator (nccl_utils.cc ftNcclInitialize over MPI in the reference) ----
        Rendezvous  rdv;
        ftcf_comm_t comm = nullptr;
        if (tp > 1) {
            rdv.dir   = rdv_dir;
            rdv.rank  = rank;
            rdv.world = world;
            (void)mkdir(rdv_dir.c_str(), 0777);
            if (exchange == "host") {
                ftcf_check(ftcf_comm_init_host_exchange(world, rank, device, &Rendezvous::allgather, &rdv, &comm), "ftcf_comm_init_host_exchange");
            }
            else {
                std::vector<uint8_t> id(FTCF_UNIQUE_ID_BYTES, 0), all((size_t)FTCF_UNIQUE_ID_BYTES * world);
                if (rank == 0) {
                    ftcf_check(ftcf_comm_get_unique_id(id.data()), "ftcf_comm_get_unique_id");
                }
                if (Rendezvous::allgather(&rdv, id.data(), all.data(), id.size()) != 0) {
                    throw std::runtime_error("rendezvous failed");
                }
                ftcf_check(ftcf_comm_init(all.data(), world, rank, device, &comm), "ftcf_comm_init");  // rank 0's id
            }
        }
        cfg.tensor_para_size = tp;
        cfg.tensor_para_rank = rank;
        cfg.pipeline_para_size = 1;
        cfg.int8_mode = int8;
        cfg.dtype = FTCF_FP16;
        cfg.use_gptj_residual = gptj ? 1 : 0;
        cfg.device = device;
        cfg.stream = stream;
        cfg.comm = comm;
        cfg.use_hip_graph = 1;
        ftcf_gptneox_weights ww{};
        ww.weights = w.data();
        ww.n_weights = (int)w.size();
        ww.int8_weights = q.empty() ? nullptr : q.data();
        ww.n_int8_weights = (int)q.size();
        ww.scales = sc.empty() ? nullptr : sc.data();
        ww.n_scales = (int)sc.size();
        ftcf_gptneox_t eng = nullptr;
        ftcf_check(ftcf_gptneox_create(&cfg, &ww, &eng), "ftcf_gptneox_create");

        const int total = (int)S + out_len;
        int*      d_ids = (int*)dev.upload(ids.data(), ids.size() * 4);
        int*      d_len = (int*)dev.upload(lens.data(), lens.size() * 4);
        int*      d_out = (int*)dev.upload(nullptr, (size_t)B * beam * total * 4);
        int*      d_seq = (int*)dev.upload(nullptr, (size_t)B * beam * 4);
        ftcf_forward_args fa{};
        fa.input_ids = d_ids;
        fa.input_lengths = d_len;
        fa.batch_size = B;
        fa.max_input_len = (int)S;
        fa.output_len = out_len;
        fa.beam_width = beam;
        fa.top_k = &top_k;
        fa.n_top_k = 1;
        fa.top_p = &top_p;
        fa.n_top_p = 1;
        fa.temperature = &temp;
        fa.n_temperature = 1;
        fa.repetition_penalty = &rep;
        fa.n_repetition_penalty = 1;
        fa.beam_search_diversity_rate = &div_rate;
        fa.n_beam_search_diversity_rate = 1;
        fa.len_penalty = &len_pen;
        fa.n_len_penalty = 1;
        fa.output_ids = d_out;
        fa.sequence_lengths = d_seq;
        ftcf_check(ftcf_gptneox_forward(eng, &fa), "ftcf_gptneox_forward");  // warm up (gptneox_example.cc:395-409)
        hipEvent_t e0, e1;
        hip_check(hipEventCreate(&e0), "event");
        hip_check(hipEventCreate(&e1), "event");
        hip_check(hipEventRecord(e0, stream), "event");
        ftcf_check(ftcf_gptneox_forward(eng, &fa), "ftcf_gptneox_forward");
        hip_check(hipEventRecord(e1, stream), "event");
        hip_check(hipEventSynchronize(e1), "event");
        float ms = 0.f;
        hip_check(hipEventElapsedTime(&ms, e0, e1), "event");

        std::vector<int> out((size_t)B * beam * total);
        hip_check(hipMemcpy(out.data(), d_out, out.size() * 4, hipMemcpyDeviceToHost), "hipMemcpy");
        if (rank == 0) {  // gptneox_example.cc:411-440: rank 0 writes, one row per (request, beam)
            std::ofstream of(out_path);
            for (size_t i = 0; i < out.size(); i++) {
                of << out[i] << " ";
                if ((i + 1) % (size_t)total == 0) {
                    of << std::endl;
                }
            }
        }
        printf("[INFO] request_batch_size %d beam_width %d head_num %d size_per_head %d total_output_len %d decoder_layers %d "
               "vocab_size %d FT-CPP-decoding-beamsearch-time %.2f ms\n",
               B, beam, nh, dh, total, L, V, ms);
        ftcf_check(ftcf_gptneox_destroy(eng), "destroy");
        if (comm) {
            ftcf_check(ftcf_comm_destroy(comm), "comm destroy");
        }
        return 0;
    }
    catch (const std::exception& e) {
        fprintf(stderr, "[ERROR] %s\n", e.what());
        return 1;
    }
}
