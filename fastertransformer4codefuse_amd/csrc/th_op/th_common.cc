// libth_common -- compiled drop-in for the reference's weight-only quantiser module
// (th_op/common/WeightOnlyQuantOps.cc:140-233 symmetric_quantize_last_axis_of_batched_matrix_int8, :344-349 the pybind11
// module; :350-356 the torch.ops registration) over the C ABI's host quantiser (include/ftcf.h ftcf_symmetric_quantize_int8).
#include <torch/extension.h>
#include <torch/library.h>

#include <string>
#include <vector>

#include "ftcf.h"

namespace th = torch;

static std::vector<th::Tensor> symmetric_quantize_last_axis_of_batched_matrix_int8(th::Tensor weight)
{
    TORCH_CHECK(!weight.is_cuda(), "weight must be a CPU tensor");  // CHECK_CPU
    TORCH_CHECK(weight.is_contiguous(), "weight must be contiguous");
    TORCH_CHECK(weight.numel() != 0, "weight should not be empty tensor");
    TORCH_CHECK(weight.dim() == 2 || weight.dim() == 3, "Invalid dim. The dim of weight should be 2 or 3");
    const auto st = weight.scalar_type();
    TORCH_CHECK(st == at::kFloat || st == at::kHalf || st == at::kBFloat16, "Invalid datatype. Weight must be FP16 or BF16");
    const size_t E = weight.dim() == 2 ? 1 : (size_t)weight.size(0);
    const size_t K = (size_t)weight.size(-2), N = (size_t)weight.size(-1);
    th::Tensor   q = th::empty_like(weight, th::dtype(th::kInt8));
    th::Tensor   scales = weight.dim() == 2 ? th::empty({(int64_t)N}, weight.options()) : th::empty({(int64_t)E, (int64_t)N}, weight.options());
    const ftcf_dtype dt = st == at::kFloat ? FTCF_FP32 : st == at::kHalf ? FTCF_FP16 : FTCF_BF16;
    if (ftcf_symmetric_quantize_int8(weight.data_ptr(), dt, E, K, N, q.data_ptr<int8_t>(), scales.data_ptr()) != 0) {
        throw std::runtime_error(std::string("[ftcf] ") + ftcf_last_error());
    }
    return {q, scales};
}

PYBIND11_MODULE(libth_common, module)
{
    module.def("symmetric_quantize_last_axis_of_batched_matrix_int8", &symmetric_quantize_last_axis_of_batched_matrix_int8,
               "symmetric_quantize_last_axis_of_batched_matrix_int8");
    module.attr("compiled") = true;
}

// (the reference's non-pybind build exposes the same function as torch.ops.fastertransformer.<name>)
TORCH_LIBRARY_FRAGMENT(fastertransformer, m)
{
    m.def("symmetric_quantize_last_axis_of_batched_matrix_int8", &symmetric_quantize_last_axis_of_batched_matrix_int8);
}
