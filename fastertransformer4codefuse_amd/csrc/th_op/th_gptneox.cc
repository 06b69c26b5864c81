// libth_gptneox -- the compiled Python extension of the engine: a thin C++ layer over the C ABI (include/ftcf.h) with the
// reference's binding surface.
//   pybind11:     libth_gptneox.GptNeoXOp(process_group, rank, head_num, ..., weights, int8_weights, scale).forward(...)
//                 (th_op/gptneox/GptNeoXOp.cc:190-212; codefuse_example.py:468-470 imports it from `lib_path`)
//   TorchScript:  torch.classes.FasterTransformer.GptNeoXOp(head_num, ..., weights, int8_weights, scale).forward(...)
//                 (th_op/gptneox/GptNeoXOp.cc:213-236), registered when the library is loaded
// The tensor-parallel communicator is bootstrapped through the CALLER's c10d::ProcessGroup (rank 0 creates the RCCL unique id,
// the group broadcasts the bytes): what nccl_inherit::ftNcclInitialize does (th_op/gptneox/utils/nccl_inherit_utils.cc:25-68)
// without reaching into ProcessGroupNCCL's protected members.
#include <torch/custom_class.h>
#include <torch/extension.h>

#include <c10/hip/HIPStream.h>
#include <pybind11/numpy.h>
#include <torch/csrc/distributed/c10d/ProcessGroup.hpp>

#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include "ftcf.h"

namespace py = pybind11;
namespace th = torch;

namespace {

void check(int rc)
{
    if (rc != 0) {
        throw std::runtime_error(std::string("[ftcf] ") + ftcf_last_error());  // (the reference prints and exit(-1)s)
    }
}

// th_utils.h:32-49 CHECK_INPUT / CHECK_TH_CUDA / CHECK_CONTIGUOUS / CHECK_TYPE
void check_input(const th::Tensor& t, const char* name)
{
    TORCH_CHECK(t.is_cuda(), name, " must be a CUDA tensor");
    TORCH_CHECK(t.is_contiguous(), name, " must be contiguous");
}
void check_input(const th::Tensor& t, const char* name, at::ScalarType st)
{
    check_input(t, name);
    TORCH_CHECK(t.scalar_type() == st, name, " has an invalid dtype");
}

// ---- tensor-parallel communicator through the caller's process group -------------------------------------------------
struct HostExchange {  // all-gather of host bytes over the group (gloo: CPU tensors), the callback of a host-exchange comm
    c10::intrusive_ptr<c10d::ProcessGroup> pg;
    int                                    world;
    static int allgather(void* user, const void* send, void* recv, size_t bytes)
    {
        auto* self = static_cast<HostExchange*>(user);
        try {
            th::Tensor src = th::empty({(int64_t)bytes}, th::kByte);
            std::memcpy(src.data_ptr(), send, bytes);
            std::vector<th::Tensor>              in{src};
            std::vector<std::vector<th::Tensor>> out(1);
            for (int r = 0; r < self->world; r++) {
                out[0].push_back(th::empty({(int64_t)bytes}, th::kByte));
            }
            self->pg->allgather(out, in)->wait();
            for (int r = 0; r < self->world; r++) {
                std::memcpy(static_cast<char*>(recv) + (size_t)r * bytes, out[0][r].data_ptr(), bytes);
            }
            return 0;
        }
        catch (const std::exception& e) {  // (an exception must not unwind through the C caller)
            fprintf(stderr, "[ftcf] host-exchange all-gather failed: %s\n", e.what());
            return 1;
        }
    }
};

ftcf_comm_t comm_from_group(const c10::intrusive_ptr<c10d::ProcessGroup>& pg, int rank, int world, int device,
                            std::shared_ptr<HostExchange>& keep)
{
    ftcf_comm_t comm = nullptr;
    const char* ex   = std::getenv("FTCF_TP_EXCHANGE");
    if (ex && std::string(ex) == "host") {
        keep = std::make_shared<HostExchange>(HostExchange{pg, world});
        check(ftcf_comm_init_host_exchange(world, rank, device, &HostExchange::allgather, keep.get(), &comm));
        return comm;
    }
    th::Tensor ids = th::zeros({FTCF_UNIQUE_ID_BYTES}, th::kByte);
    if (rank == 0) {
        check(ftcf_comm_get_unique_id(ids.data_ptr<uint8_t>()));
    }
    const bool              on_device = pg->getBackendName() == "nccl";
    th::Tensor              t         = on_device ? ids.to(th::Device(th::kCUDA, device)) : ids;
    std::vector<th::Tensor> v{t};
    c10d::BroadcastOptions  opt;
    opt.rootRank = 0;
    pg->broadcast(v, opt)->wait();
    ids = v[0].cpu().contiguous();
    check(ftcf_comm_init(ids.data_ptr<uint8_t>(), world, rank, device, &comm));
    return comm;
}

// ---- the engine behind both bindings (th_op/gptneox/GptNeoXOp.h:69-231 FTGptNeoX ctor, :246-381 forward) ------------
class Engine {
public:
    Engine(ftcf_comm_t comm, int64_t rank, int64_t head_num, int64_t size_per_head, int64_t inter_size, int64_t layer_num,
           int64_t vocab_size, int64_t rotary_embedding_dim, int64_t start_id, int64_t end_id, int64_t tensor_para_size,
           int64_t pipeline_para_size, int64_t int8_mode, int64_t /*max_seq_len*/, bool use_gptj_residual,
           std::vector<th::Tensor> weights, std::vector<th::Tensor> int8_weights, std::vector<th::Tensor> scale):
        comm_(comm), weights_(std::move(weights)), int8_weights_(std::move(int8_weights)), scale_(std::move(scale))
    {
        TORCH_CHECK(!weights_.empty(), "weights must not be empty");
        const auto st = weights_[0].scalar_type();  // GptNeoXOp.cc:46: the dtype of weights[0] selects the engine
        for (size_t i = 0; i < weights_.size(); i++) {
            check_input(weights_[i], "weights");
            TORCH_CHECK(weights_[i].numel() == 0 || weights_[i].scalar_type() == st,
                        "Invalid datatype. All weights must have the same dtype");
        }
        TORCH_CHECK(st == at::kHalf || st == at::kFloat, "Wrong Tensor type.");  // GptNeoXOp.cc:56-105
        TORCH_CHECK(!(st == at::kFloat && int8_mode != 0), "int8_mode needs half weights");
        for (const auto& t : int8_weights_) {
            check_input(t, "int8_weights", at::kChar);
        }
        for (const auto& t : scale_) {
            check_input(t, "scale", at::kHalf);
        }
        device_ = weights_[0].device().index();
        auto ptrs = [](const std::vector<th::Tensor>& ts) {
            std::vector<const void*> v(ts.size() ? ts.size() : 1, nullptr);
            for (size_t i = 0; i < ts.size(); i++) {
                v[i] = ts[i].numel() > 0 ? ts[i].data_ptr() : nullptr;
            }
            return v;
        };
        const auto           wp = ptrs(weights_), qp = ptrs(int8_weights_), sp = ptrs(scale_);
        ftcf_gptneox_weights w{wp.data(), (int)weights_.size(), qp.data(), (int)int8_weights_.size(), sp.data(), (int)scale_.size()};
        ftcf_gptneox_config  c{};
        c.head_num             = (int)head_num;
        c.size_per_head        = (int)size_per_head;
        c.inter_size           = (int)inter_size;
        c.num_layer            = (int)layer_num;
        c.vocab_size           = (int)vocab_size;
        c.rotary_embedding_dim = (int)rotary_embedding_dim;
        c.start_id             = (int)start_id;
        c.end_id               = (int)end_id;
        c.tensor_para_size     = (int)tensor_para_size;
        c.tensor_para_rank     = (int)(rank % tensor_para_size);
        c.pipeline_para_size   = (int)pipeline_para_size;
        c.int8_mode            = (int)int8_mode;
        c.dtype                = st == at::kFloat ? FTCF_FP32 : FTCF_FP16;
        c.use_gptj_residual    = use_gptj_residual ? 1 : 0;
        c.device               = device_;
        c.stream               = c10::hip::getCurrentHIPStream(device_).stream();  // GptNeoXOp.h:180-185
        c.comm                 = comm_;
        c.use_hip_graph        = 1;
        check(ftcf_gptneox_create(&c, &w, &h_));
    }
    ~Engine()
    {
        if (h_) {
            ftcf_gptneox_destroy(h_);
        }
        if (comm_) {
            ftcf_comm_destroy(comm_);
        }
    }
    Engine(const Engine&) = delete;
    Engine& operator=(const Engine&) = delete;

    // GptNeoXOp::forward (GptNeoXOp.cc:113-185)
    std::vector<th::Tensor> forward(th::Tensor input_ids, th::Tensor input_lengths, int64_t output_len,
                                    th::optional<int64_t> beam_width_opt, th::optional<th::Tensor> top_k_opt,
                                    th::optional<th::Tensor> top_p_opt, th::optional<th::Tensor> beam_search_diversity_rate_opt,
                                    th::optional<th::Tensor> temperature_opt, th::optional<th::Tensor> len_penalty_opt,
                                    th::optional<th::Tensor> repetition_penalty_opt, th::optional<th::Tensor> random_seed_opt,
                                    th::optional<th::Tensor> stop_words_list_opt, th::optional<th::Tensor> optional_last_tokens_opt,
                                    th::optional<int64_t> return_cum_log_probs_opt, ftcf_token_callback cb, void* cb_user,
                                    th::optional<th::Tensor> debug_logits, bool release_gil)
    {
        check_input(input_ids, "input_ids");
        TORCH_CHECK(input_ids.scalar_type() == at::kInt, "input_ids dtype should be int32");
        TORCH_CHECK(input_ids.dim() == 2, "input_ids must be a matrix");
        check_input(input_lengths, "input_lengths");
        TORCH_CHECK(input_lengths.scalar_type() == at::kInt, "input_lengths dtype should be int32");
        TORCH_CHECK(input_lengths.dim() == 1, "input_lengths must be a vector");
        const int64_t return_cum_log_probs = return_cum_log_probs_opt.has_value() ? return_cum_log_probs_opt.value() : 0;
        TORCH_CHECK(return_cum_log_probs == 0 || return_cum_log_probs == 1,
                    "return_cum_log_probs should be 0 (no return cum_log_probs), 1 (the cumulative log probs of generated "
                    "sequences)");
        const int  beam_width = beam_width_opt.has_value() ? (int)beam_width_opt.value() : 1;
        const int  B = (int)input_ids.size(0), S = (int)input_ids.size(1);
        TORCH_CHECK(input_lengths.numel() == B, "input_lengths must hold one length per row of input_ids");
        const auto i32 = th::dtype(th::kInt32).device(input_ids.device()).requires_grad(false);
        th::Tensor output_ids       = th::empty({B, beam_width, S + output_len}, i32);
        th::Tensor sequence_lengths = th::empty({B, beam_width}, i32);
        th::Tensor cum_log_probs    = th::empty({B, beam_width}, i32.dtype(th::kFloat32));

        std::vector<th::Tensor> keep;  // host copies of the runtime arguments, alive until the call returns
        auto host = [&](const th::optional<th::Tensor>& t, at::ScalarType st, const char* name, const void*& p, int& n) {
            p = nullptr;
            n = 0;
            if (t.has_value()) {
                TORCH_CHECK(!t.value().is_cuda(), name, " must be a CPU tensor");
                keep.push_back(t.value().to(st).contiguous().view({-1}));
                p = keep.back().data_ptr();
                n = (int)keep.back().numel();
            }
        };
        ftcf_forward_args a{};
        a.input_ids     = input_ids.data_ptr<int>();
        a.input_lengths = input_lengths.data_ptr<int>();
        a.batch_size    = B;
        a.max_input_len = S;
        a.output_len    = (int)output_len;
        a.beam_width    = beam_width;
        const void* p   = nullptr;
        host(top_k_opt, at::kInt, "top_k", p, a.n_top_k);
        a.top_k = static_cast<const int*>(p);
        host(top_p_opt, at::kFloat, "top_p", p, a.n_top_p);
        a.top_p = static_cast<const float*>(p);
        host(beam_search_diversity_rate_opt, at::kFloat, "beam_search_diversity_rate", p, a.n_beam_search_diversity_rate);
        a.beam_search_diversity_rate = static_cast<const float*>(p);
        host(temperature_opt, at::kFloat, "temperature", p, a.n_temperature);
        a.temperature = static_cast<const float*>(p);
        host(len_penalty_opt, at::kFloat, "len_penalty", p, a.n_len_penalty);
        a.len_penalty = static_cast<const float*>(p);
        host(repetition_penalty_opt, at::kFloat, "repetition_penalty", p, a.n_repetition_penalty);
        a.repetition_penalty = static_cast<const float*>(p);
        host(random_seed_opt, at::kLong, "random_seed", p, a.n_random_seed);  // (int64 bits = the uint64 seed)
        a.random_seed = static_cast<const uint64_t*>(p);
        if (stop_words_list_opt.has_value()) {
            check_input(stop_words_list_opt.value(), "stop_words_list", at::kInt);
            TORCH_CHECK(stop_words_list_opt.value().dim() == 3 && stop_words_list_opt.value().size(0) == B
                            && stop_words_list_opt.value().size(1) == 2,
                        "stop_words_list must have shape [batch_size, 2, stop_words_length]");
            a.stop_words_list = stop_words_list_opt.value().data_ptr<int>();
            a.stop_words_len  = (int)stop_words_list_opt.value().size(2);
        }
        if (optional_last_tokens_opt.has_value()) {
            check_input(optional_last_tokens_opt.value(), "optional_last_tokens", at::kInt);
            TORCH_CHECK(optional_last_tokens_opt.value().dim() == 2 && optional_last_tokens_opt.value().size(0) == B,
                        "optional_last_tokens must have shape [batch_size, count]");
            a.optional_last_tokens       = optional_last_tokens_opt.value().data_ptr<int>();
            a.optional_last_tokens_count = (int)optional_last_tokens_opt.value().size(1);
        }
        a.return_cum_log_probs = (int)return_cum_log_probs;
        a.callback             = cb;
        a.callback_user        = cb_user;
        a.output_ids           = output_ids.data_ptr<int>();
        a.sequence_lengths     = sequence_lengths.data_ptr<int>();
        a.cum_log_probs        = return_cum_log_probs ? cum_log_probs.data_ptr<float>() : nullptr;
        if (debug_logits.has_value()) {
            check_input(debug_logits.value(), "_debug_logits", at::kFloat);
            a.debug_logits = debug_logits.value().data_ptr<float>();
        }
        int rc = 0;
        if (release_gil) {
            py::gil_scoped_release nogil;  // (the token callback takes it back)
            rc = ftcf_gptneox_forward(h_, &a);
        }
        else {
            rc = ftcf_gptneox_forward(h_, &a);
        }
        check(rc);
        if (return_cum_log_probs > 0) {
            return {output_ids, sequence_lengths, cum_log_probs};
        }
        return {output_ids, sequence_lengths};
    }
    py::dict stats() const
    {
        ftcf_forward_stats s{};
        check(ftcf_gptneox_get_stats(h_, &s));
        py::dict d;
        d["prefill_ms"]    = s.prefill_ms;
        d["decode_ms"]     = s.decode_ms;
        d["decode_steps"]  = s.decode_steps;
        d["gemv_ms_sum"]   = s.gemv_ms_sum;
        d["gemv_launches"] = s.gemv_launches;
        d["gemv_bytes"]    = s.gemv_bytes;
        d["gemv_kind"]     = s.gemv_kind;
        d["decode_path"]   = s.decode_path;
        d["prefill_overlap"]       = s.prefill_overlap;
        d["prefill_ms_plain"]      = s.prefill_ms_plain;
        d["prefill_ms_overlapped"] = s.prefill_ms_overlapped;
        d["window_allreduces"]     = s.window_allreduces;
        d["persist_layout"]        = s.persist_layout;
        d["decode_overlap"]        = s.decode_overlap;
        d["decode_step_ms_plain"]  = s.decode_step_ms_plain;
        d["decode_step_ms_overlapped"] = s.decode_step_ms_overlapped;
        return d;
    }

private:
    ftcf_gptneox_t          h_    = nullptr;
    ftcf_comm_t             comm_ = nullptr;
    int                     device_ = 0;
    std::vector<th::Tensor> weights_, int8_weights_, scale_;  // kept alive, like GptNeoXOp.h:402-404
};

// ---- pybind11 face ---------------------------------------------------------------------------------------------------
class PyGptNeoXOp {
public:
    PyGptNeoXOp(py::object group, int64_t rank, int64_t head_num, int64_t size_per_head, int64_t inter_size, int64_t layer_num,
                int64_t vocab_size, int64_t rotary_embedding_dim, int64_t start_id, int64_t end_id, int64_t tensor_para_size,
                int64_t pipeline_para_size, int64_t int8_mode, int64_t max_seq_len, bool use_gptj_residual,
                std::vector<th::Tensor> weights, std::vector<th::Tensor> int8_weights, std::vector<th::Tensor> scale)
    {
        TORCH_CHECK(!weights.empty(), "weights must not be empty");
        check_input(weights[0], "weights[0]");
        const int   device = weights[0].device().index();
        ftcf_comm_t comm   = nullptr;
        if (tensor_para_size > 1) {
            const int tp_rank = (int)(rank % tensor_para_size);
            const char* fake  = std::getenv("FTCF_FAKE_TP");
            if (fake && std::string(fake) == "1") {
                // timing aid (bench.py --fake-tp N): one rank of a TP = N job over a 1-rank communicator
                uint8_t id[FTCF_UNIQUE_ID_BYTES];
                check(ftcf_comm_get_unique_id(id));
                check(ftcf_comm_init(id, 1, 0, device, &comm));
            }
            else if (py::hasattr(group, "id") && !py::isinstance<py::str>(group)) {
                // LocalTensorParallelGroup (test infrastructure): the ranks inside one process on one device
                auto ids = group.attr("id").cast<py::array_t<uint8_t>>();
                TORCH_CHECK(ids.size() == FTCF_UNIQUE_ID_BYTES, "bad local group id");
                check(ftcf_comm_init_local(ids.data(), (int)tensor_para_size, tp_rank, device, &comm));
            }
            else {
                // the caller's torch.distributed group (th_op/gptneox/GptNeoXOp.cc:26 `const c10d::ProcessGroup&`)
                auto pg = group.cast<c10::intrusive_ptr<c10d::ProcessGroup>>();
                comm    = comm_from_group(pg, tp_rank, (int)tensor_para_size, device, hx_);
            }
        }
        eng_ = std::make_unique<Engine>(comm, rank, head_num, size_per_head, inter_size, layer_num, vocab_size,
                                        rotary_embedding_dim, start_id, end_id, tensor_para_size, pipeline_para_size, int8_mode,
                                        max_seq_len, use_gptj_residual, std::move(weights), std::move(int8_weights),
                                        std::move(scale));
    }
    std::vector<th::Tensor> forward(th::Tensor input_ids, th::Tensor input_lengths, int64_t output_len,
                                    th::optional<int64_t> beam_width, th::optional<th::Tensor> top_k,
                                    th::optional<th::Tensor> top_p, th::optional<th::Tensor> beam_search_diversity_rate,
                                    th::optional<th::Tensor> temperature, th::optional<th::Tensor> len_penalty,
                                    th::optional<th::Tensor> repetition_penalty, th::optional<th::Tensor> random_seed,
                                    th::optional<th::Tensor> stop_words_list, th::optional<th::Tensor> optional_last_tokens,
                                    th::optional<int64_t> return_cum_log_probs, py::object callback,
                                    th::optional<th::Tensor> debug_logits)
    {
        // pybind_callback_utils.cc:22-103: {"last_tokens": [[...] * beam] * batch, "idxs": ...} after every step but the last
        struct Cb {
            py::object fn;
            static void call(const int* tokens, const int* idxs, int batch, int beam, void* user)
            {
                py::gil_scoped_acquire gil;
                auto*                  self = static_cast<Cb*>(user);
                py::list               lt, ix;
                for (int b = 0; b < batch; b++) {
                    py::list l1, l2;
                    for (int w = 0; w < beam; w++) {
                        l1.append(tokens[b * beam + w]);
                        l2.append(idxs[b * beam + w]);
                    }
                    lt.append(l1);
                    ix.append(l2);
                }
                py::dict d;
                d["last_tokens"] = lt;
                d["idxs"]        = ix;
                self->fn(d);
            }
        } cb{callback};
        const bool has_cb = !callback.is_none();
        return eng_->forward(input_ids, input_lengths, output_len, beam_width, top_k, top_p, beam_search_diversity_rate, temperature,
                             len_penalty, repetition_penalty, random_seed, stop_words_list, optional_last_tokens,
                             return_cum_log_probs, has_cb ? &Cb::call : nullptr, has_cb ? &cb : nullptr, debug_logits, true);
    }
    py::dict stats() const { return eng_->stats(); }

private:
    std::shared_ptr<HostExchange> hx_;
    std::unique_ptr<Engine>       eng_;
};

// ---- TorchScript face (GptNeoXOp.cc:213-236: no process group, no callback) -------------------------------------------
class ThsGptNeoXOp: public th::CustomClassHolder {
public:
    ThsGptNeoXOp(int64_t head_num, int64_t size_per_head, int64_t inter_size, int64_t layer_num, int64_t vocab_size,
                 int64_t rotary_embedding_dim, int64_t start_id, int64_t end_id, int64_t tensor_para_size,
                 int64_t pipeline_para_size, int64_t int8_mode, int64_t max_seq_len, bool use_gptj_residual,
                 std::vector<th::Tensor> weights, std::vector<th::Tensor> int8_weights, std::vector<th::Tensor> scale)
    {
        TORCH_CHECK(tensor_para_size == 1, "the TorchScript class carries no process group: tensor_para_size must be 1 "
                                           "(use libth_gptneox.GptNeoXOp for tensor parallelism)");
        eng_ = std::make_unique<Engine>(nullptr, 0, head_num, size_per_head, inter_size, layer_num, vocab_size,
                                        rotary_embedding_dim, start_id, end_id, tensor_para_size, pipeline_para_size, int8_mode,
                                        max_seq_len, use_gptj_residual, std::move(weights), std::move(int8_weights),
                                        std::move(scale));
    }
    std::vector<th::Tensor> forward(th::Tensor input_ids, th::Tensor input_lengths, int64_t output_len,
                                    th::optional<int64_t> beam_width, th::optional<th::Tensor> top_k,
                                    th::optional<th::Tensor> top_p, th::optional<th::Tensor> beam_search_diversity_rate,
                                    th::optional<th::Tensor> temperature, th::optional<th::Tensor> len_penalty,
                                    th::optional<th::Tensor> repetition_penalty, th::optional<th::Tensor> random_seed,
                                    th::optional<th::Tensor> stop_words_list, th::optional<th::Tensor> optional_last_tokens,
                                    th::optional<int64_t> return_cum_log_probs)
    {
        return eng_->forward(input_ids, input_lengths, output_len, beam_width, top_k, top_p, beam_search_diversity_rate, temperature,
                             len_penalty, repetition_penalty, random_seed, stop_words_list, optional_last_tokens,
                             return_cum_log_probs, nullptr, nullptr, th::nullopt, false);
    }

private:
    std::unique_ptr<Engine> eng_;
};

}  // namespace

PYBIND11_MODULE(libth_gptneox, module)
{
    module.doc() = "compiled drop-in for the reference's libth_gptneox (th_op/gptneox/GptNeoXOp.cc:190-212) over libftcf.so";
    py::class_<PyGptNeoXOp>(module, "GptNeoXOp")
        .def(py::init<py::object, int64_t, int64_t, int64_t, int64_t, int64_t, int64_t, int64_t, int64_t, int64_t, int64_t,
                      int64_t, int64_t, int64_t, bool, std::vector<th::Tensor>, std::vector<th::Tensor>, std::vector<th::Tensor>>())
        .def("forward", &PyGptNeoXOp::forward, py::arg("input_ids"), py::arg("input_lengths"), py::arg("output_len"),
             py::arg("beam_width") = py::none(), py::arg("top_k") = py::none(), py::arg("top_p") = py::none(),
             py::arg("beam_search_diversity_rate") = py::none(), py::arg("temperature") = py::none(),
             py::arg("len_penalty") = py::none(), py::arg("repetition_penalty") = py::none(),
             py::arg("random_seed") = py::none(), py::arg("stop_words_list") = py::none(),
             py::arg("optional_last_tokens") = py::none(), py::arg("return_cum_log_probs") = py::none(),
             py::arg("callback") = py::none(), py::arg("_debug_logits") = py::none())
        .def("stats", &PyGptNeoXOp::stats);
    module.attr("compiled") = true;
}

static auto ftcf_gptneox_ths =
    th::class_<ThsGptNeoXOp>("FasterTransformer", "GptNeoXOp")
        .def(th::init<int64_t, int64_t, int64_t, int64_t, int64_t, int64_t, int64_t, int64_t, int64_t, int64_t, int64_t, int64_t,
                      bool, std::vector<th::Tensor>, std::vector<th::Tensor>, std::vector<th::Tensor>>())
        .def("forward", &ThsGptNeoXOp::forward);
