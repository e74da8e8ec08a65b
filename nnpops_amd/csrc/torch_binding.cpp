// torch_binding.cpp -- the PyTorch operator surface of the reference, re-implemented on top of the
// C ABI (include/nnpops_hip.h).  Plain C++ (no kernels here): every entry point hands raw device
// pointers of contiguous tensors to libnnpops_hip.so on the current HIP stream.
//
// Registrations mirror the reference one-for-one so that TorchScript modules and user code written
// against NNPOps keep working:
//   torch.classes.NNPOpsANISymmetryFunctions.Holder / torch.ops.NNPOpsANISymmetryFunctions.operation
//                                                         (reference src/pytorch/SymmetryFunctions.cpp:265-284)
//   torch.classes.NNPOpsCFConvNeighbors.Holder            (reference src/pytorch/CFConvNeighbors.cpp:77-85)
//   torch.classes.NNPOpsCFConv.Holder / torch.ops.NNPOpsCFConv.operation
//                                                         (reference src/pytorch/CFConv.cpp:276-291)
//   torch.ops.neighbors.getNeighborPairs                  (reference src/pytorch/neighbors/neighbors.cpp:4)
//   torch.ops.NNPOpsBatchedNN.BatchedLinear               (reference src/pytorch/BatchedNN.cpp:48-50)
//
// Differences that are deliberate:
//   * the AEV / CFConv / BatchedNN ops have no CPU implementation: a CPU tensor raises (the reference's CPU path is the
//     oracle of this repository, not part of the product).  The two ops for which the reference itself registers a CPU
//     kernel at the dispatcher -- neighbors::getNeighborPairs and pme::pme_direct -- do have a CPU key here (plain C++
//     loops pinned to the reference's CPU ops by fixtures, tests/test_neighbors_cpu_key.py, tests/test_pme_cpu.py);
//   * outputs are fresh tensors on every call (the reference re-returns the same storage,
//     SymmetryFunctions.cpp:136-138,157) -- no caller can observe the difference except by aliasing bugs;
//   * CFConv honours the current stream (the reference leaves that commented out, CFConv.cpp:167-170).
#include <c10/hip/HIPGuard.h>
#include <c10/hip/HIPStream.h>
#include <hip/hip_runtime_api.h>
#include <torch/script.h>
#include <torch/serialize/archive.h>

#include <cmath>
#include <cstdlib>
#include <limits>
#include <mutex>
#include <sstream>
#include <stdexcept>
#include <string>
#include <vector>

#include "../../include/nnpops_hip.h"

// hash of the sources this binding was compiled from (torch_binding.py embeds it and refuses a stale binary)
#ifndef NNPOPS_BINDING_HASH
#define NNPOPS_BINDING_HASH "unknown"
#endif
extern "C" __attribute__((used, visibility("default"))) const char nnpops_torch_binding_version[] =
    "nnpops_torch_binding src:" NNPOPS_BINDING_HASH;

namespace {

using torch::Tensor;
using torch::autograd::AutogradContext;
using torch::autograd::tensor_list;

[[noreturn]] void raise_last(const char* what) {
    throw std::runtime_error(std::string(what) + ": " + nnpops_last_error());
}

void* current_stream(const torch::Device& device) {
    return (void*)c10::hip::getCurrentHIPStream(device.index()).stream();
}

bool stream_is_capturing(void* stream) {
    hipStreamCaptureStatus status = hipStreamCaptureStatusNone;
    if (hipStreamIsCapturing((hipStream_t)stream, &status) != hipSuccess) return false;
    return status != hipStreamCaptureStatusNone;
}

// ---------------------------------------------------------------------------------------------
// The atomic networks of a frame on nnpops_mlp_forward / nnpops_mlp_input_grad (mlp_fused.hip).  The packed parameters
// travel as two flat buffers (what nnpops_amd/BatchedNN.py::_FusedSpeciesNN registers): per kind, one after the other,
//   planes (fp16): w0 (M members) | w2 (M) | w4 (M) | w4t (M) | w2t (M) | w0t          floats: b0 | b2 | b4 | w6 | b6
// with the packed widths h1, h2, h3 of every kind in `widths` (multiples of 32).  With x_blocks (the 16-column blocks of x
// the networks are packed over, nnpops_hip.h: x_groups) the first layer's planes are that narrow, and when they hold at most
// 256 columns every kind carries one more set, w0tm (M), behind w0t: the forward launch then forms the input gradient itself.
// ---------------------------------------------------------------------------------------------
struct MlpCall {
    nnpops_mlp_frame frame{};
    Tensor energies;                  // [atoms][members]
    std::vector<Tensor> keep;         // workspaces
};
// workspaces of a frame that a caller with a home for them (the AEV holder of the one-node step) keeps between the steps
struct MlpScratch {
    Tensor partial, energies;
};

int64_t mlp_halves(int64_t rows, int64_t cols) { return nnpops_mlp_packed_halves((int)rows, (int)cols); }

MlpCall mlp_prepare(const Tensor& x, const Tensor& rows, const std::vector<int64_t>& kind_atoms, const std::vector<int64_t>& widths,
                    int64_t members, const Tensor& planes, const Tensor& floats, bool with_gradient,
                    const c10::optional<Tensor>& x_blocks = c10::nullopt, const c10::optional<Tensor>& dead_blocks = c10::nullopt,
                    int64_t act_scale_log2 = 4, MlpScratch* scratch = nullptr) {
    TORCH_CHECK(act_scale_log2 >= 4 && act_scale_log2 <= 12, "act_scale_log2: 4..12 (activations are scaled by 2^-k before the fp16 split)");
    TORCH_CHECK(x.is_cuda() && x.dim() == 2 && x.scalar_type() == torch::kFloat32 && x.is_contiguous(),
                "the fused networks take a contiguous [atoms, features] float32 device tensor");
    const bool narrow = x_blocks.has_value() && x_blocks->numel() > 0;
    const int64_t kinds = (int64_t)kind_atoms.size(), atoms = x.size(0), F = narrow ? 16 * x_blocks->numel() : x.size(1);
    if (narrow) {
        TORCH_CHECK(x.size(1) % 16 == 0 && F <= x.size(1), "x_blocks: blocks of 16 columns of a [atoms, multiple of 16] array");
        for (const Tensor* t : {&*x_blocks, dead_blocks.has_value() ? &*dead_blocks : &*x_blocks})
            TORCH_CHECK(t->scalar_type() == torch::kInt32 && t->is_contiguous() && t->device() == x.device(), "column block lists must be int32 on the device of x");
        TORCH_CHECK(dead_blocks.has_value() && x_blocks->numel() + dead_blocks->numel() == x.size(1) / 16,
                    "x_blocks and dead_blocks together must list every 16-column block of x once");
    }
    const bool in_forward = narrow && with_gradient && F <= 256;
    TORCH_CHECK(kinds >= 1 && kinds <= NNPOPS_MLP_MAX_KINDS && (int64_t)widths.size() == 3 * kinds, "1..", NNPOPS_MLP_MAX_KINDS, " kinds, three widths each");
    TORCH_CHECK(rows.scalar_type() == torch::kInt32 && rows.is_contiguous() && rows.device() == x.device() && rows.numel() == atoms,
                "rows must be an int32 permutation of the atoms on the device of x");
    TORCH_CHECK(planes.scalar_type() == torch::kFloat16 && planes.is_contiguous() && planes.device() == x.device() &&
                floats.scalar_type() == torch::kFloat32 && floats.is_contiguous() && floats.device() == x.device(),
                "packed network parameters must be contiguous fp16 / fp32 buffers on the device of x");
    MlpCall c;
    nnpops_mlp_frame& fr = c.frame;
    fr.num_kinds = (int)kinds; fr.num_features = (int)F; fr.num_members = (int)members;
    fr.x = x.data_ptr<float>(); fr.ldx = (int)x.size(1); fr.rows = rows.data_ptr<int32_t>(); fr.alpha = 0.1f; fr.act_scale_log2 = (int)act_scale_log2;       // BatchedNN.py:103
    if (narrow) {
        fr.x_groups = x_blocks->data_ptr<int32_t>();
        fr.dead_groups = dead_blocks->numel() ? dead_blocks->data_ptr<int32_t>() : nullptr;
        fr.num_dead_groups = (int)dead_blocks->numel();
    }
    auto fresh = [&](Tensor& kept, at::IntArrayRef shape) {      // (a kept workspace of the right shape on the right device, or a new one)
        if (!kept.defined() || kept.sizes() != shape || kept.device() != x.device()) kept = torch::empty(shape, x.options());
        return kept;
    };
    if (in_forward) {
        Tensor local;
        Tensor partial = fresh(scratch ? scratch->partial : local, {members, atoms, F});
        fr.dx_partial = partial.data_ptr<float>();
        c.keep.push_back(partial);
    }
    {
        Tensor local;
        c.energies = fresh(scratch ? scratch->energies : local, {atoms, members});
    }
    fr.energies = c.energies.data_ptr<float>();
    const at::Half* ph = planes.data_ptr<at::Half>();
    const float* pf = floats.data_ptr<float>();
    int64_t oh = 0, of = 0, total = 0;
    for (int64_t k = 0; k < kinds; k++) {
        const int64_t h1 = widths[3 * k], h2 = widths[3 * k + 1], h3 = widths[3 * k + 2], M = members;
        nnpops_mlp_kind& kd = fr.kinds[k];
        kd.num_atoms = (int)kind_atoms[k]; kd.h1 = (int)h1; kd.h2 = (int)h2; kd.h3 = (int)h3;
        total += kind_atoms[k];
        kd.w0 = ph + oh;  oh += M * mlp_halves(h1, F);
        kd.w2 = ph + oh;  oh += M * mlp_halves(h2, h1);
        kd.w4 = ph + oh;  oh += M * mlp_halves(h3, h2);
        kd.w4t = ph + oh; oh += M * mlp_halves(h2, h3);
        kd.w2t = ph + oh; oh += M * mlp_halves(h1, h2);
        kd.w0t = ph + oh; oh += mlp_halves(F, M * h1);
        if (narrow && F <= 256) { kd.w0tm = ph + oh; oh += M * mlp_halves(F, h1); }
        kd.b0 = pf + of; of += M * h1;
        kd.b2 = pf + of; of += M * h2;
        kd.b4 = pf + of; of += M * h3;
        kd.w6 = pf + of; of += M * h3;
        kd.b6 = pf + of; of += M;
        if (with_gradient && !in_forward) {
            Tensor d1 = torch::empty({std::max<int64_t>(nnpops_mlp_d1_halves((int)kind_atoms[k], (int)M, (int)h1), 1)}, planes.options());
            kd.d1 = d1.data_ptr();
            c.keep.push_back(d1);
        }
    }
    TORCH_CHECK(total == atoms, "the kinds hold ", total, " atoms, x has ", atoms);
    TORCH_CHECK(oh == planes.numel() && of == floats.numel(), "packed network parameters do not match the widths (", oh, " / ", planes.numel(),
                " fp16 values, ", of, " / ", floats.numel(), " floats)");
    return c;
}

void require_device_tensor(const Tensor& t, const char* name) {
    if (!t.is_cuda())
        throw std::runtime_error(std::string("Unsupported device for \"") + name + "\": " + t.device().str() +
                                 " (this build of NNPOps runs on AMD GPUs only; there is no CPU path)");
}

}  // namespace

// =============================================================================================
// ANI symmetry functions
// =============================================================================================
namespace NNPOps {
namespace ANISymmetryFunctions {

class Holder;
using HolderPtr = torch::intrusive_ptr<Holder>;

class Holder : public torch::CustomClassHolder {
public:
    Holder(int64_t numSpecies, double Rcr, double Rca, const std::vector<double>& EtaR, const std::vector<double>& ShfR,
           const std::vector<double>& EtaA, const std::vector<double>& Zeta, const std::vector<double>& ShfA,
           const std::vector<double>& ShfZ, const std::vector<int64_t>& atomSpecies)
        : numSpecies(numSpecies), Rcr(Rcr), Rca(Rca), EtaR(EtaR), ShfR(ShfR), EtaA(EtaA), Zeta(Zeta), ShfA(ShfA), ShfZ(ShfZ),
          atomSpecies(atomSpecies) {}

    ~Holder() override {
        if (impl) nnpops_ani_destroy(impl);
    }

    // Additive: how often forward() pays the host round trip that verifies the neighbour capacities (and grows them).
    // 1 (default) = every call, k = every k-th call, 0 = only the first.  Between checks an overflow goes unnoticed, as
    // inside a captured graph -- for production loops whose densities are known (cf. getNeighborPairs' checkErrors).
    // Extension: the builders' sticky overflow word as an int32[1] DEVICE tensor (no copy, no synchronisation): non-zero when a
    // forward since the last capacity check overflowed a neighbour buffer -- what a caller replaying a captured graph, where no
    // check can run, looks at (nnpops_hip.h: nnpops_ani_overflow_word).  Needs one forward() first.
    Tensor overflowFlag() {
        if (!impl) throw std::runtime_error("overflow_flag() called before forward()");
        const int32_t* word = nullptr;
        if (nnpops_ani_overflow_word(impl, &word) != NNPOPS_OK) raise_last("NNPOpsANISymmetryFunctions::overflow_flag");
        // (the tensor keeps this Holder -- and with it the handle that owns the word -- alive: ADVICE r04; the word itself is cleared by
        //  every capacity check, so "sticky" means between two checks)
        auto self = c10::intrusive_ptr<Holder>::unsafe_reclaim_from_nonowning(this);
        return torch::from_blob(const_cast<int32_t*>(word), {1}, [self](void*) {}, torch::TensorOptions().device(device).dtype(torch::kInt32));
    }

    void setCheckInterval(int64_t interval) {
        if (interval < 0) throw std::runtime_error("The check interval has to be >= 0");
        checkInterval = interval;
    }

    // Additive (the reference has no batch dimension, SymmetryFunctions.py:110): the atoms of this Holder
    // are `offsets.size() - 1` independent NON-PERIODIC molecules, molecule m = atoms [offsets[m], offsets[m+1]); atoms of
    // different molecules never see each other, forward/backward evaluate the whole batch in one launch sequence
    // (nnpops_ani_set_molecules).  An empty list restores the single-system behaviour.
    void setMolecules(const std::vector<int64_t>& offsets) {
        if (!offsets.empty() && (offsets.front() != 0 || offsets.back() != (int64_t)atomSpecies.size()))
            throw std::runtime_error("molecule offsets have to start at 0 and end at the number of atoms");
        moleculeOffsets = offsets;
        if (impl) applyMolecules();
    }

    tensor_list forward(const Tensor& positions, const c10::optional<Tensor>& cellOpt) { return forwardImpl(positions, cellOpt, false); }

    // fused = true: ONE output, aev [N, S*nR + S(S+1)/2*nA] = (radial | angular) per row -- what TorchANI's AEVComputer
    // returns and what the Python wrapper otherwise builds with torch.cat: the kernels write the two parts in place
    // (nnpops_ani_compute_strided), no concatenation copy forward, no split copy backward.
    // defer_check: a caller with more launches to queue behind the AEV (EnergyFunction) takes the capacity check in two halves --
    // the copy of the overflow word is queued here, finishDeferredCheck() reads it after those launches (nnpops_hip.h:
    // nnpops_ani_check_begin / _end), so the host round trip no longer stops the device between the AEV and its consumers.
    tensor_list forwardImpl(const Tensor& positions, const c10::optional<Tensor>& cellOpt, bool fused, bool defer_check = false) {
        // same checks, same messages as the reference (SymmetryFunctions.cpp:76-99)
        if (positions.scalar_type() != torch::kFloat32) throw std::runtime_error("The type of \"positions\" has to be float32");
        if (positions.dim() != 2) throw std::runtime_error("The shape of \"positions\" has to have 2 dimensions");
        if (positions.size(0) != (int64_t)atomSpecies.size())
            throw std::runtime_error("The size of the 1nd dimension of \"positions\" has to be " + std::to_string(atomSpecies.size()));
        if (positions.size(1) != 3) throw std::runtime_error("The size of the 2nd dimension of \"positions\" has to be 3");
        require_device_tensor(positions, "positions");
        Tensor cell;
        if (cellOpt) {
            cell = *cellOpt;
            if (cell.scalar_type() != torch::kFloat32) throw std::runtime_error("The type of \"cell\" has to be float32");
            if (cell.dim() != 2) throw std::runtime_error("The shape of \"cell\" has to have 2 dimensions");
            if (cell.size(0) != 3) throw std::runtime_error("The size of the 1nd dimension of \"cell\" has to be 3");
            if (cell.size(1) != 3) throw std::runtime_error("The size of the 2nd dimension of \"cell\" has to be 3");
            if (cell.device() != positions.device()) throw std::runtime_error("\"cell\" has to be on the same device as \"positions\"");
            cell = cell.contiguous();
        }
        if (!impl) {
            device = positions.device();
            periodic = cellOpt.has_value();       // frozen at the first call, like the reference (:122)
            std::vector<float> radial, angular;
            for (double eta : EtaR)
                for (double rs : ShfR) { radial.push_back((float)eta); radial.push_back((float)rs); }              // :110-113
            for (double eta : EtaA)
                for (double zeta : Zeta)
                    for (double rs : ShfA)
                        for (double thetas : ShfZ) {                                                              // :115-120
                            angular.push_back((float)eta); angular.push_back((float)rs);
                            angular.push_back((float)zeta); angular.push_back((float)thetas);
                        }
            std::vector<int32_t> species(atomSpecies.begin(), atomSpecies.end());
            if (nnpops_ani_create(&impl, (int)species.size(), (int)numSpecies, (float)Rcr, (float)Rca, periodic ? 1 : 0,
                                  species.data(), (int)(radial.size() / 2), radial.data(), (int)(angular.size() / 4),
                                  angular.data(), /*torchani=*/1, device.index()) != NNPOPS_OK)
                raise_last("NNPOpsANISymmetryFunctions");
            numRadial = (int64_t)(radial.size() / 2);
            numAngular = (int64_t)(angular.size() / 4);
            if (!moleculeOffsets.empty()) applyMolecules();
        }
        if (positions.device() != device) throw std::runtime_error("The device of \"positions\" has changed");
        if (periodic && !cellOpt) throw std::runtime_error("\"cell\" is required: this Holder was first used with periodic box vectors");

        const Tensor pos = positions.contiguous();
        const int64_t n = (int64_t)atomSpecies.size();
        const auto opts = torch::TensorOptions().device(device).dtype(torch::kFloat32);
        const int64_t wr = numSpecies * numRadial, wa = numSpecies * (numSpecies + 1) / 2 * numAngular;
        Tensor radial, angular, aev;
        float *pr, *pa;
        int ld = 0;
        if (fused) {
            aev = torch::empty({n, wr + wa}, opts);
            pr = aev.data_ptr<float>(); pa = pr + wr; ld = (int)(wr + wa);
        } else {
            radial = torch::empty({n, wr}, opts);
            angular = torch::empty({n, wa}, opts);
            pr = radial.data_ptr<float>(); pa = angular.data_ptr<float>();
        }
        void* stream = current_stream(device);
        nnpops_ani_set_stream(impl, stream);
        const bool capturing = stream_is_capturing(stream);
        for (int attempt = 0;; attempt++) {
            if (nnpops_ani_compute_strided(impl, pos.data_ptr<float>(), periodic ? cell.data_ptr<float>() : nullptr, pr, ld, pa, ld) !=
                NNPOPS_OK)
                raise_last("NNPOpsANISymmetryFunctions::forward");
            if (capturing) break;                     // no host synchronisation inside a graph capture
            // (additive knob, like getNeighborPairs' checkErrors: the capacity check costs a host round trip;
            // interval k checks every k-th call, 0 never again after the first)
            // (forceCheck: this is the re-issue of a step whose deferred check reported an overflow -- a second overflow, e.g. row
            //  capacity after the cell bins, must be caught now, whatever the interval; the re-issue is not a counted call)
            const bool due = forceCheck || calls == 0 || (checkInterval > 0 && calls % checkInterval == 0);
            if (!forceCheck) calls++;
            forceCheck = false;
            if (!due && attempt == 0) break;
            if (defer_check && publishInline) {                // (the caller's next launch publishes the word: energy_step)
                if (nnpops_ani_check_begin_with(impl, &publishWord, &publishTo, &publishStamp) == 1) { checkPending = true; break; }
                publishWord = nullptr;
            } else if (defer_check && nnpops_ani_check_begin(impl) == 1) { checkPending = true; break; }
            const int rc = nnpops_ani_check(impl, nullptr, nullptr);
            if (rc == NNPOPS_OK) break;
            if (rc != NNPOPS_ERR_CAPACITY || attempt > 8) raise_last("NNPOpsANISymmetryFunctions::forward");
        }
        if (fused) return {aev};
        return {radial, angular};
    }

    tensor_list backwardFused(const Tensor& aevGrad) {
        if (!impl) throw std::runtime_error("backward() called before forward()");
        const Tensor g = aevGrad.contiguous();
        const int64_t wr = numSpecies * numRadial, w = g.size(1);
        Tensor positionsGrad = torch::empty({(int64_t)atomSpecies.size(), 3},
                                            torch::TensorOptions().device(device).dtype(torch::kFloat32));
        nnpops_ani_set_stream(impl, current_stream(device));
        if (nnpops_ani_backprop_strided(impl, g.data_ptr<float>(), (int)w, g.data_ptr<float>() + wr, (int)w,
                                        positionsGrad.data_ptr<float>()) != NNPOPS_OK)
            raise_last("NNPOpsANISymmetryFunctions::backward");
        return {Tensor(), positionsGrad, Tensor()};
    }

    tensor_list backward(const tensor_list& grads) {
        if (!impl) throw std::runtime_error("backward() called before forward()");
        const Tensor radialGrad = grads[0].contiguous();     // the reference clones to force a dense buffer (:162-163)
        const Tensor angularGrad = grads[1].contiguous();
        Tensor positionsGrad = torch::empty({(int64_t)atomSpecies.size(), 3},
                                            torch::TensorOptions().device(device).dtype(torch::kFloat32));
        nnpops_ani_set_stream(impl, current_stream(device));
        if (nnpops_ani_backprop(impl, radialGrad.data_ptr<float>(), angularGrad.data_ptr<float>(),
                                positionsGrad.data_ptr<float>()) != NNPOPS_OK)
            raise_last("NNPOpsANISymmetryFunctions::backward");
        return {Tensor(), positionsGrad, Tensor()};          // no gradient for the holder and the box (:174)
    }

    static std::string serialize(const HolderPtr& self) {
        torch::serialize::OutputArchive archive;
        archive.write("numSpecies", self->numSpecies);
        archive.write("Rcr", self->Rcr);
        archive.write("Rca", self->Rca);
        archive.write("EtaR", self->EtaR);
        archive.write("ShfR", self->ShfR);
        archive.write("EtaA", self->EtaA);
        archive.write("Zeta", self->Zeta);
        archive.write("ShfA", self->ShfA);
        archive.write("ShfZ", self->ShfZ);
        archive.write("atomSpecies", self->atomSpecies);
        std::stringstream stream;
        archive.save_to(stream);
        return stream.str();
    }

    // -> true: the neighbour buffers had overflowed and have been grown; the caller issues forwardImpl() and its consumers again
    bool finishDeferredCheck() {
        if (!checkPending) return false;
        checkPending = false;
        const int rc = nnpops_ani_check_end(impl);
        if (rc == NNPOPS_OK) return false;
        if (rc != NNPOPS_ERR_CAPACITY) raise_last("NNPOpsANISymmetryFunctions::forward");
        forceCheck = true;              // the caller issues the step again: that build is verified whatever the check interval
        return true;
    }

    static HolderPtr deserialize(const std::string& state) {
        std::stringstream stream(state);
        torch::serialize::InputArchive archive;
        archive.load_from(stream, torch::kCPU);
        torch::IValue numSpecies, Rcr, Rca, EtaR, ShfR, EtaA, Zeta, ShfA, ShfZ, atomSpecies;
        archive.read("numSpecies", numSpecies);
        archive.read("Rcr", Rcr);
        archive.read("Rca", Rca);
        archive.read("EtaR", EtaR);
        archive.read("ShfR", ShfR);
        archive.read("EtaA", EtaA);
        archive.read("Zeta", Zeta);
        archive.read("ShfA", ShfA);
        archive.read("ShfZ", ShfZ);
        archive.read("atomSpecies", atomSpecies);
        return HolderPtr::make(numSpecies.toInt(), Rcr.toDouble(), Rca.toDouble(), EtaR.toDoubleVector(), ShfR.toDoubleVector(),
                               EtaA.toDoubleVector(), Zeta.toDoubleVector(), ShfA.toDoubleVector(), ShfZ.toDoubleVector(),
                               atomSpecies.toIntVector());
    }

private:
    void applyMolecules() {
        std::vector<int32_t> off(moleculeOffsets.begin(), moleculeOffsets.end());
        if (nnpops_ani_set_molecules(impl, off.empty() ? 0 : (int)off.size() - 1, off.empty() ? nullptr : off.data()) != NNPOPS_OK)
            raise_last("NNPOpsANISymmetryFunctions::set_molecules");
    }

    int64_t numSpecies;
    double Rcr, Rca;
    std::vector<double> EtaR, ShfR, EtaA, Zeta, ShfA, ShfZ;
    std::vector<int64_t> atomSpecies;
    std::vector<int64_t> moleculeOffsets;   // additive: batched molecules (setMolecules)
    torch::Device device = torch::kCPU;
    bool periodic = false;
    int64_t numRadial = 0, numAngular = 0;
    nnpops_ani_t impl = nullptr;
    int64_t checkInterval = 1;      // capacity check every k-th forward (0: only the first); see setCheckInterval
    bool forceCheck = false;        // the next forwardImpl() re-issues a step after a reported overflow: always checked
    int64_t calls = 0;
    bool checkPending = false;
public:
    // energy_step: the capacity check's word is published by nnpops_mlp_forward (frame.publish_*) instead of a launch of its own
    bool publishInline = false;
    const int32_t* publishWord = nullptr; int32_t* publishTo = nullptr; int32_t publishStamp = 0;
    // energy_step: dE/dAEV of the frame, kept between the steps.  The networks write only the column blocks the molecule's species
    // can fill (x_blocks); the others are zero and stay zero -- cleared once, when the buffer is made, instead of by every step
    // (7 of the 8 MB a step of the 2 001-atom water box used to write there).  Valid for one list of live blocks.
    Tensor gradCache;
    const void* gradCacheBlocks = nullptr;
    Tensor gradCacheKey;              // the list of live blocks itself, held: while the cache is valid its address cannot be freed and handed to
    uint32_t gradCacheVersion = 0;    //   another list, and an in-place edit of the list shows in its version counter
    MlpScratch mlpScratch;            // ... and the networks' workspaces of a step (per-member gradient shares, per-atom energies)
    void resetGradientCache() { gradCache = Tensor(); gradCacheBlocks = nullptr; gradCacheKey = Tensor(); }      // (the networks' live blocks have changed: BatchedNN.py)
private:
};

class AutogradFunctions : public torch::autograd::Function<AutogradFunctions> {
public:
    static tensor_list forward(AutogradContext* ctx, const HolderPtr& holder, const Tensor& positions,
                               const c10::optional<Tensor>& periodicBoxVectors) {
        ctx->saved_data["holder"] = holder;
        return holder->forward(positions, periodicBoxVectors);
    }
    static tensor_list backward(AutogradContext* ctx, const tensor_list& grads) {
        const auto holder = ctx->saved_data["holder"].toCustomClass<Holder>();
        ctx->saved_data.erase("holder");
        return holder->backward(grads);
    }
};

tensor_list operation(const c10::optional<HolderPtr>& holder, const Tensor& positions,
                      const c10::optional<Tensor>& periodicBoxVectors) {
    return AutogradFunctions::apply(*holder, positions, periodicBoxVectors);
}

class FusedAutogradFunction : public torch::autograd::Function<FusedAutogradFunction> {
public:
    static Tensor forward(AutogradContext* ctx, const HolderPtr& holder, const Tensor& positions,
                          const c10::optional<Tensor>& periodicBoxVectors) {
        ctx->saved_data["holder"] = holder;
        return holder->forwardImpl(positions, periodicBoxVectors, true)[0];
    }
    static tensor_list backward(AutogradContext* ctx, const tensor_list& grads) {
        const auto holder = ctx->saved_data["holder"].toCustomClass<Holder>();
        ctx->saved_data.erase("holder");
        return holder->backwardFused(grads[0]);
    }
};

// Additive: the whole AEV as one tensor (see Holder::forwardImpl).  operation() keeps the reference's two-tensor form.
Tensor aev(const c10::optional<HolderPtr>& holder, const Tensor& positions, const c10::optional<Tensor>& periodicBoxVectors) {
    return FusedAutogradFunction::apply(*holder, positions, periodicBoxVectors);
}

// ---------------------------------------------------------------------------------------------
// Additive: the whole OptimizedTorchANI step (reference OptimizedTorchANI.py:49-52: aev_computer -> neural_networks) as ONE
// autograd node.  forward: AEV kernels -> the fused networks (mlp_fused.hip) -> ensemble-mean energy; when the positions
// require a gradient the same call runs the networks' input-gradient pass and the AEV backward and keeps dE/dpositions, so
// that backward() is one multiplication -- no autograd graph over the ~20 small tensor ops the four-module composition
// records, no [N, 1008] gradient held by autograd between the passes.  (The reference's PME op keeps its derivatives
// the same way, pmeCPU.cpp:161-171.)
// ---------------------------------------------------------------------------------------------
// One energy (+ gradient) evaluation of the frame: AEV forward, networks, and -- with a gradient -- the networks' input gradient and
// the AEV backward.  Returns {energy [1], dE/dpositions * gradient_sign [N, 3] or undefined}.  (The capacity check of the AEV
// holder -- one host round trip per call unless set_check_interval says otherwise -- is taken in two halves: its word is published
// right behind the AEV forward, read after everything else has been launched.  A buffer that did overflow is grown there and the
// step issued again.  Before: the host waited for the AEV forward, then launched the networks into an idle device; waiting at the
// END of the step for the whole stream was worse still, 0.22 -> 0.28 ms.)
std::pair<Tensor, Tensor> energy_step(const HolderPtr& holder, const Tensor& frame, const c10::optional<Tensor>& cell, const Tensor& rows,
                                      const std::vector<int64_t>& kind_atoms, const std::vector<int64_t>& widths, int64_t members,
                                      const Tensor& planes, const Tensor& floats, const c10::optional<Tensor>& shift,
                                      const c10::optional<Tensor>& x_blocks, const c10::optional<Tensor>& dead_blocks, bool need_gradient,
                                      float gradient_sign, int64_t act_scale_log2) {
    TORCH_CHECK(frame.dim() == 2 || (frame.dim() == 3 && frame.size(0) == 1), "energy(): positions must be [atoms, 3] or [1, atoms, 3]");
    const Tensor positions = frame.dim() == 3 ? frame[0] : frame;
    Tensor energy, kept;
    for (int attempt = 0;; attempt++) {
        TORCH_CHECK(attempt <= 8, "NNPOpsANISymmetryFunctions::energy: neighbour buffers kept overflowing");
        holder->publishInline = true; holder->publishWord = nullptr;
        const Tensor aev = holder->forwardImpl(positions, cell, true, /*defer_check=*/true)[0];
        holder->publishInline = false;
        c10::hip::HIPGuard guard(aev.device().index());
        void* stream = current_stream(aev.device());
        MlpCall call = mlp_prepare(aev, rows, kind_atoms, widths, members, planes, floats, need_gradient, x_blocks, dead_blocks, act_scale_log2,
                                   &holder->mlpScratch);
        if (holder->publishWord) {                              // the deferred check's word goes out with the first network launch
            call.frame.publish_word = holder->publishWord; call.frame.publish_to = holder->publishTo; call.frame.publish_stamp = holder->publishStamp;
        }
        if (nnpops_mlp_forward(stream, &call.frame, need_gradient ? 1 : 0) != NNPOPS_OK) raise_last("NNPOpsANISymmetryFunctions::energy");
        // the ensemble mean (BatchedNN.py:109), shifted by the self energy when the caller hands it over: its own small launch,
        // or -- when the input gradient is only a sum over the members (dx_partial) -- a passenger of that launch
        const bool mean_rides = need_gradient && call.frame.dx_partial != nullptr;
        if (shift.has_value()) {
            TORCH_CHECK(shift->scalar_type() == torch::kFloat64 && shift->device() == aev.device() && shift->numel() == 1 && shift->is_contiguous(),
                        "energy(): the self-energy shift must be one float64 on the device of the positions");
            energy = torch::empty({1}, aev.options().dtype(torch::kFloat64));
        } else {
            energy = torch::empty({1}, aev.options());
        }
        const float mean_scale = 1.0f / (float)members;
        if (mean_rides) {
            call.frame.mean_scale = mean_scale;
            if (shift.has_value()) { call.frame.mean_shift = shift->data_ptr<double>(); call.frame.mean_out_shifted = energy.data_ptr<double>(); }
            else call.frame.mean_out = energy.data_ptr<float>();
        } else {
            const int rc = shift.has_value()
                ? nnpops_mlp_energy_mean_shifted(stream, call.energies.data_ptr<float>(), call.energies.numel(), mean_scale,
                                                 shift->data_ptr<double>(), energy.data_ptr<double>())
                : nnpops_mlp_energy_mean(stream, call.energies.data_ptr<float>(), call.energies.numel(), mean_scale, energy.data_ptr<float>());
            if (rc != NNPOPS_OK) raise_last("NNPOpsANISymmetryFunctions::energy");
        }
        if (need_gradient) {
            Tensor daev;
            if (call.frame.dx_partial != nullptr && x_blocks.has_value()) {     // (the sum over the members writes the live blocks only)
                if (!holder->gradCache.defined() || holder->gradCache.sizes() != aev.sizes() || holder->gradCache.device() != aev.device() ||
                    holder->gradCacheBlocks != x_blocks->data_ptr() || holder->gradCacheVersion != x_blocks->_version()) {
                    holder->gradCache = torch::zeros_like(aev);
                    holder->gradCacheBlocks = x_blocks->data_ptr();
                    holder->gradCacheKey = *x_blocks;
                    holder->gradCacheVersion = x_blocks->_version();
                }
                daev = holder->gradCache;
                call.frame.num_dead_groups = 0;
            } else {
                daev = torch::empty_like(aev);
            }
            call.frame.dx = daev.data_ptr<float>(); call.frame.lddx = (int)daev.size(1); call.frame.dx_scale = gradient_sign / (float)members;
            if (nnpops_mlp_input_grad(stream, &call.frame) != NNPOPS_OK) raise_last("NNPOpsANISymmetryFunctions::energy");
            kept = holder->backwardFused(daev)[1];
        }
        if (!holder->finishDeferredCheck()) break;
    }
    return {energy, kept};
}

class EnergyFunction : public torch::autograd::Function<EnergyFunction> {
public:
    // `frame`: positions [N, 3], or [1, N, 3] as the torchani modules pass them (the gradient comes back in the same shape: no
    // select / select-backward kernels around the node).  `shift`: optional float64 device scalar, the molecule's self energy
    // (EnergyShifter.py:52); with it the energy comes back in double precision, promoted and shifted as the reference does it.
    static Tensor forward(AutogradContext* ctx, const HolderPtr& holder, const Tensor& frame, const c10::optional<Tensor>& cell,
                          const Tensor& rows, std::vector<int64_t> kind_atoms, std::vector<int64_t> widths, int64_t members,
                          const Tensor& planes, const Tensor& floats, const c10::optional<Tensor>& shift,
                          const c10::optional<Tensor>& x_blocks, const c10::optional<Tensor>& dead_blocks, bool need_gradient,
                          int64_t act_scale_log2) {
        Tensor energy, kept;
        std::tie(energy, kept) = energy_step(holder, frame, cell, rows, kind_atoms, widths, members, planes, floats, shift, x_blocks, dead_blocks,
                                             need_gradient, 1.0f, act_scale_log2);
        if (need_gradient) {
            ctx->save_for_backward({kept});
            ctx->saved_data["lead"] = frame.dim() == 3;
        }
        return energy;
    }
    static tensor_list backward(AutogradContext* ctx, const tensor_list& grads) {
        TORCH_CHECK(!torch::GradMode::is_enabled(),
                    "NNPOpsANISymmetryFunctions::energy: second derivatives are not implemented (backward was called with create_graph=True); "
                    "use the four-module composition for that");
        const auto saved = ctx->get_saved_variables();
        TORCH_CHECK(!saved.empty(), "energy() was evaluated without a gradient request");
        const Tensor& kept = saved[0];                                       // dE/dpositions, [N, 3] float32
        const Tensor g = grads[0].contiguous();
        TORCH_CHECK(g.numel() == 1 && g.device() == kept.device() && (g.scalar_type() == torch::kFloat32 || g.scalar_type() == torch::kFloat64),
                    "energy(): unexpected gradient of the energy");
        c10::hip::HIPGuard guard(kept.device().index());
        Tensor out = torch::empty_like(kept);
        if (nnpops_scale_by_scalar(current_stream(kept.device()), kept.data_ptr<float>(), kept.numel(), g.data_ptr(),
                                   g.scalar_type() == torch::kFloat64 ? 1 : 0, out.data_ptr<float>()) != NNPOPS_OK)
            raise_last("NNPOpsANISymmetryFunctions::energy (backward)");
        if (ctx->saved_data["lead"].toBool()) out = out.unsqueeze(0);
        return {Tensor(), out, Tensor(), Tensor(), Tensor(), Tensor(), Tensor(), Tensor(), Tensor(), Tensor(), Tensor(), Tensor(), Tensor(), Tensor()};
    }
};

Tensor energy(const c10::optional<HolderPtr>& holder, const Tensor& positions, const c10::optional<Tensor>& cell, const Tensor& rows,
              std::vector<int64_t> kind_atoms, std::vector<int64_t> widths, int64_t members, const Tensor& planes, const Tensor& floats,
              const c10::optional<Tensor>& shift, const c10::optional<Tensor>& x_blocks, const c10::optional<Tensor>& dead_blocks,
              int64_t act_scale_log2) {
    const bool need = torch::GradMode::is_enabled() && positions.requires_grad();
    return EnergyFunction::apply(*holder, positions, cell, rows, kind_atoms, widths, members, planes, floats, shift, x_blocks, dead_blocks, need,
                                 act_scale_log2);
}

// Additive: energy AND forces (-dE/dpositions) of the frame from one call, outside autograd -- what an MD driver asks a model for
// when it takes the forces as an output instead of differentiating the energy (the step is the one of energy() with a gradient:
// the same launches; no autograd node, no sum / ones / scale kernels around it).
std::tuple<Tensor, Tensor> energy_forces(const c10::optional<HolderPtr>& holder, const Tensor& positions, const c10::optional<Tensor>& cell,
                                         const Tensor& rows, std::vector<int64_t> kind_atoms, std::vector<int64_t> widths, int64_t members,
                                         const Tensor& planes, const Tensor& floats, const c10::optional<Tensor>& shift,
                                         const c10::optional<Tensor>& x_blocks, const c10::optional<Tensor>& dead_blocks,
                                         int64_t act_scale_log2) {
    torch::NoGradGuard no_grad;
    Tensor energy, forces;
    std::tie(energy, forces) = energy_step(*holder, positions.detach(), cell, rows, kind_atoms, widths, members, planes, floats, shift, x_blocks,
                                           dead_blocks, true, -1.0f, act_scale_log2);
    if (positions.dim() == 3) forces = forces.unsqueeze(0);
    return std::make_tuple(energy, forces);
}

TORCH_LIBRARY(NNPOpsANISymmetryFunctions, m) {
    m.class_<Holder>("Holder")
        .def(torch::init<int64_t, double, double, const std::vector<double>&, const std::vector<double>&,
                         const std::vector<double>&, const std::vector<double>&, const std::vector<double>&,
                         const std::vector<double>&, const std::vector<int64_t>&>())
        .def("forward", &Holder::forward)
        .def("backward", &Holder::backward)
        .def("set_check_interval", &Holder::setCheckInterval)
        .def("overflow_flag", &Holder::overflowFlag)
        .def("set_molecules", &Holder::setMolecules)
        .def("reset_gradient_cache", &Holder::resetGradientCache)
        .def_pickle([](const HolderPtr& self) -> std::string { return Holder::serialize(self); },
                    [](const std::string& state) -> HolderPtr { return Holder::deserialize(state); });
    m.def("operation", operation);
    m.def("aev", aev);
    m.def("energy(__torch__.torch.classes.NNPOpsANISymmetryFunctions.Holder? holder, Tensor positions, Tensor? cell, Tensor rows, int[] kind_atoms, "
          "int[] widths, int members, Tensor planes, Tensor floats, Tensor? shift=None, Tensor? x_blocks=None, Tensor? dead_blocks=None, int act_scale_log2=4) -> Tensor", energy);
    m.def("energy_forces(__torch__.torch.classes.NNPOpsANISymmetryFunctions.Holder? holder, Tensor positions, Tensor? cell, Tensor rows, int[] kind_atoms, "
          "int[] widths, int members, Tensor planes, Tensor floats, Tensor? shift=None, Tensor? x_blocks=None, Tensor? dead_blocks=None, int act_scale_log2=4) -> (Tensor, Tensor)",
          energy_forces);
}

}  // namespace ANISymmetryFunctions

// =============================================================================================
// CFConv neighbours
// =============================================================================================
namespace CFConvNeighbors {

class Holder : public torch::CustomClassHolder {
public:
    explicit Holder(double cutoff) : cutoff(cutoff) {}
    ~Holder() override {
        if (impl) nnpops_cfconv_neighbors_destroy(impl);
    }

    void build(const Tensor& positions) { buildImpl(positions, nullptr); }

    // Additive to the reference (whose binding is non-periodic, CFConvNeighbors.cpp:52,57,74, although its core
    // supports a box, CFConv.h:57): the same list under periodic boundary conditions.  box: (3, 3), rows = vectors.
    void buildPeriodic(const Tensor& positions, const Tensor& box) {
        if (box.scalar_type() != torch::kFloat32) throw std::runtime_error("The type of \"box\" has to be float32");
        if (box.dim() != 2 || box.size(0) != 3 || box.size(1) != 3) throw std::runtime_error("The shape of \"box\" has to be (3, 3)");
        require_device_tensor(box, "box");
        const Tensor b = box.detach().contiguous();
        buildImpl(positions, &b);
    }

    void buildImpl(const Tensor& positions, const Tensor* box) {
        if (positions.scalar_type() != torch::kFloat32) throw std::runtime_error("The type of \"positions\" has to be float32");
        if (positions.dim() != 2) throw std::runtime_error("The shape of \"positions\" has to have 2 dimensions");
        if (positions.size(1) != 3) throw std::runtime_error("The size of the 2nd dimension of \"positions\" has to be 3");
        require_device_tensor(positions, "positions");
        if (!impl) {
            numAtoms = positions.size(0);
            device = positions.device();
            periodic = box != nullptr;
            if (nnpops_cfconv_neighbors_create(&impl, (int)numAtoms, (float)cutoff, periodic ? 1 : 0, device.index()) != NNPOPS_OK)
                raise_last("NNPOpsCFConvNeighbors");
        }
        if (positions.size(0) != numAtoms) throw std::runtime_error("The size of the 2nd dimension of \"positions\" has changed");
        if (positions.device() != device) throw std::runtime_error("The device of \"positions\" has changed");
        if (periodic != (box != nullptr)) throw std::runtime_error("The periodicity of \"neighbors\" has changed");
        if (box && box->device() != device) throw std::runtime_error("The device of \"box\" has changed");
        const Tensor pos = positions.detach().contiguous();
        void* stream = current_stream(device);
        nnpops_cfconv_neighbors_set_stream(impl, stream);
        const bool capturing = stream_is_capturing(stream);
        for (int attempt = 0;; attempt++) {
            if (nnpops_cfconv_neighbors_build(impl, pos.data_ptr<float>(), box ? box->data_ptr<float>() : nullptr) != NNPOPS_OK)
                raise_last("CFConvNeighbors::build");
            if (capturing) break;
            const int rc = nnpops_cfconv_neighbors_check(impl, nullptr);
            if (rc == NNPOPS_OK) break;
            if (rc != NNPOPS_ERR_CAPACITY || attempt > 8) raise_last("CFConvNeighbors::build");
        }
    }
    double getCutoff() const { return cutoff; }
    nnpops_cfconv_neighbors_t getImpl() const { return impl; }

private:
    double cutoff;
    int64_t numAtoms = 0;
    bool periodic = false;
    torch::Device device = torch::kCPU;
    nnpops_cfconv_neighbors_t impl = nullptr;
};
using HolderPtr = torch::intrusive_ptr<Holder>;

TORCH_LIBRARY(NNPOpsCFConvNeighbors, m) {
    m.class_<Holder>("Holder")
        .def(torch::init<double>())
        .def("build", &Holder::build)
        .def("build_periodic", &Holder::buildPeriodic)
        .def_pickle([](const HolderPtr& self) -> double { return self->getCutoff(); },
                    [](double cutoff) -> HolderPtr { return HolderPtr::make(cutoff); });
}

}  // namespace CFConvNeighbors

// =============================================================================================
// CFConv
// =============================================================================================
namespace CFConv {

using Neighbors = NNPOps::CFConvNeighbors::Holder;
using NeighborsPtr = torch::intrusive_ptr<Neighbors>;
class Holder;
using HolderPtr = torch::intrusive_ptr<Holder>;

class Holder : public torch::CustomClassHolder {
public:
    Holder(double gaussianWidth, const std::string& activation, const Tensor& weights1, const Tensor& biases1,
           const Tensor& weights2, const Tensor& biases2)
        : gaussianWidth(gaussianWidth), activation(activation),
          // host copies, as in the reference (CFConv.cpp:63-66)
          weights1(weights1.to(torch::kFloat32).cpu().clone()), biases1(biases1.to(torch::kFloat32).cpu().clone()),
          weights2(weights2.to(torch::kFloat32).cpu().clone()), biases2(biases2.to(torch::kFloat32).cpu().clone()) {}

    ~Holder() override {
        if (impl) nnpops_cfconv_destroy(impl);
    }

    Tensor forward(const c10::IValue& neighbors_, const Tensor& positions_, const Tensor& input_) {
        neighbors = neighbors_.toCustomClass<Neighbors>();      // kept for the backward pass
        if (positions_.scalar_type() != torch::kFloat32) throw std::runtime_error("The type of \"positions\" has to be float32");
        if (positions_.dim() != 2) throw std::runtime_error("The shape of \"positions\" has to have 2 dimensions");
        if (positions_.size(1) != 3) throw std::runtime_error("The size of the 2nd dimension of \"positions\" has to be 3");
        if (input_.device() != positions_.device()) throw std::runtime_error("The device of \"input\" and \"positions\" has to be the same");
        if (input_.scalar_type() != torch::kFloat32) throw std::runtime_error("The type of \"input\" has to be float32");
        if (input_.dim() != 2) throw std::runtime_error("The shape of \"input\" has to have 2 dimensions");
        if (input_.size(0) != positions_.size(0))
            throw std::runtime_error("The size of the 1nd dimension of \"input\" has to be equal to the 1st dimension of \"positions\"");
        require_device_tensor(positions_, "positions");
        positions = positions_.detach().contiguous();
        input = input_.detach().contiguous();
        if (!impl) {
            device = positions.device();
            numAtoms = positions.size(0);
            numFilters = input.size(1);
            cutoff = neighbors->getCutoff();
            int act;
            if (activation == "ssp") act = 0;
            else if (activation == "tanh") act = 1;
            else throw std::invalid_argument("Invalid value of \"activation\"");
            // shape checks and messages of the reference (CFConv.cpp:106-127)
            if (weights1.dim() != 2) throw std::runtime_error("The shape of \"weights1\" has to have 2 dimensions");
            const int64_t numGaussians = weights1.size(0);
            if (weights1.size(1) != numFilters)
                throw std::runtime_error("The size of the 2nd dimension of \"weights1\" has to be equal to the 2st dimension of \"input\"");
            if (biases1.dim() != 1) throw std::runtime_error("The shape of \"biases1\" has to have 1 dimension");
            if (biases1.size(0) != numFilters) throw std::runtime_error("The size of \"biases1\" has to be equal to the 2st dimension of \"input\"");
            if (weights2.dim() != 2) throw std::runtime_error("The shape of \"weights2\" has to have 2 dimensions");
            if (weights2.size(0) != numFilters)
                throw std::runtime_error("The size of the 1nd dimension of \"weights2\" has to be equal to the 2st dimension of \"input\"");
            if (weights2.size(1) != numFilters)
                throw std::runtime_error("The size of the 2nd dimension of \"weights2\" has to be equal to the 2st dimension of \"input\"");
            if (biases2.dim() != 1) throw std::runtime_error("The shape of \"biases2\" has to have 1 dimension");
            if (biases2.size(0) != numFilters) throw std::runtime_error("The size of \"biases2\" has to be equal to the 2st dimension of \"input\"");
            // Layout note: the reference hands the contiguous [G, W] buffer of weights1 to a core that indexes it
            // as [W][G] (CFConv.cpp:131-132 -> CpuCFConv.cpp:163), i.e. a reinterpretation, not a transpose.  The
            // C ABI takes the core layout, so the same buffer is passed through unchanged.
            const Tensor w1 = weights1.contiguous(), w2 = weights2.contiguous();
            if (nnpops_cfconv_create(&impl, (int)numAtoms, (int)numFilters, (int)numGaussians, (float)cutoff, 0, (float)gaussianWidth,
                                     act, w1.data_ptr<float>(), biases1.data_ptr<float>(), w2.data_ptr<float>(),
                                     biases2.data_ptr<float>(), device.index()) != NNPOPS_OK)
                raise_last("NNPOpsCFConv");
        }
        if (neighbors->getCutoff() != cutoff) throw std::runtime_error("The cutoff of \"neighbors\" has changed");
        if (positions.size(0) != numAtoms) throw std::runtime_error("The size of the 1nd dimension of \"positions\" has changed");
        if (positions.device() != device) throw std::runtime_error("The device of \"positions\" has changed");
        if (input.size(0) != numAtoms) throw std::runtime_error("The size of the 1nd dimension of \"input\" has changed");
        if (input.size(1) != numFilters) throw std::runtime_error("The size of the 2nd dimension of \"input\" has changed");
        if (input.device() != device) throw std::runtime_error("The device of \"input\" has changed");
        if (!neighbors->getImpl()) throw std::runtime_error("\"neighbors\" has not been built");

        Tensor output = torch::empty({numAtoms, numFilters}, torch::TensorOptions().device(device).dtype(torch::kFloat32));
        nnpops_cfconv_set_stream(impl, current_stream(device));
        if (nnpops_cfconv_compute(impl, neighbors->getImpl(), positions.data_ptr<float>(), nullptr, input.data_ptr<float>(),
                                  output.data_ptr<float>()) != NNPOPS_OK)
            raise_last("NNPOpsCFConv::forward");
        return output;
    }

    tensor_list backward(const tensor_list& grads) {
        if (!impl) throw std::runtime_error("backward() called before forward()");
        const Tensor outputGrad = grads[0].contiguous();
        const auto opts = torch::TensorOptions().device(device).dtype(torch::kFloat32);
        Tensor inputGrad = torch::empty({numAtoms, numFilters}, opts);
        Tensor positionsGrad = torch::empty({numAtoms, 3}, opts);
        nnpops_cfconv_set_stream(impl, current_stream(device));
        if (nnpops_cfconv_backprop(impl, neighbors->getImpl(), positions.data_ptr<float>(), nullptr, input.data_ptr<float>(),
                                   outputGrad.data_ptr<float>(), inputGrad.data_ptr<float>(), positionsGrad.data_ptr<float>()) != NNPOPS_OK)
            raise_last("NNPOpsCFConv::backward");
        return {Tensor(), Tensor(), positionsGrad, inputGrad};    // nothing for the holder and the neighbours (:189)
    }

    static std::string serialize(const HolderPtr& self) {
        torch::serialize::OutputArchive archive;
        archive.write("gaussianWidth", self->gaussianWidth);
        archive.write("activation", self->activation);
        archive.write("weights1", self->weights1);
        archive.write("biases1", self->biases1);
        archive.write("weights2", self->weights2);
        archive.write("biases2", self->biases2);
        std::stringstream stream;
        archive.save_to(stream);
        return stream.str();
    }

    static HolderPtr deserialize(const std::string& state) {
        std::stringstream stream(state);
        torch::serialize::InputArchive archive;
        archive.load_from(stream, torch::kCPU);
        torch::IValue gaussianWidth, activation;
        Tensor weights1, biases1, weights2, biases2;
        archive.read("gaussianWidth", gaussianWidth);
        archive.read("activation", activation);
        archive.read("weights1", weights1);
        archive.read("biases1", biases1);
        archive.read("weights2", weights2);
        archive.read("biases2", biases2);
        return HolderPtr::make(gaussianWidth.toDouble(), activation.toStringRef(), weights1, biases1, weights2, biases2);
    }

private:
    double gaussianWidth;
    std::string activation;
    Tensor weights1, biases1, weights2, biases2;
    torch::Device device = torch::kCPU;
    int64_t numAtoms = 0, numFilters = 0;
    double cutoff = 0;
    NeighborsPtr neighbors;
    Tensor positions, input;
    nnpops_cfconv_t impl = nullptr;
};

class AutogradFunctions : public torch::autograd::Function<AutogradFunctions> {
public:
    static Tensor forward(AutogradContext* ctx, const HolderPtr& holder, const c10::IValue& neighbors, const Tensor& positions,
                          const Tensor& input) {
        ctx->saved_data["holder"] = holder;
        return holder->forward(neighbors, positions, input);
    }
    static tensor_list backward(AutogradContext* ctx, const tensor_list& grads) {
        const HolderPtr holder = ctx->saved_data["holder"].toCustomClass<Holder>();
        ctx->saved_data.erase("holder");
        return holder->backward(grads);
    }
};

Tensor operation(const c10::optional<HolderPtr>& holder, const c10::IValue& neighbors, const Tensor& positions,
                 const Tensor& input) {
    return AutogradFunctions::apply(*holder, neighbors, positions, input);
}

TORCH_LIBRARY(NNPOpsCFConv, m) {
    m.class_<Holder>("Holder")
        .def(torch::init<double, const std::string&, const Tensor&, const Tensor&, const Tensor&, const Tensor&>())
        .def("forward", &Holder::forward)
        .def("backward", &Holder::backward)
        .def_pickle([](const HolderPtr& self) -> std::string { return Holder::serialize(self); },
                    [](const std::string& state) -> HolderPtr { return Holder::deserialize(state); });
    m.def("operation", operation);
}

}  // namespace CFConv
}  // namespace NNPOps

// =============================================================================================
// getNeighborPairs
// =============================================================================================
namespace {

// The transposed index of the LAST list getNeighborPairs emitted on a device with an index (round 6): pme::pme_direct, the list's
// immediate consumer (src/pytorch/pme/pme.py:163-165), finds it here when it is handed that very list -- same storage, same version
// counter, held alive so that the address cannot be handed to another tensor -- and then runs without atomics
// (nnpops_pme_direct_indexed); any other list (edited, shuffled, built by the caller) takes the entry point that assumes nothing.
struct PairIndexCache {
    Tensor neighbors, index;
    uint32_t version = 0;
};
PairIndexCache& pair_index_cache(int device) {
    // (never destroyed: tensors released from a static destructor would reach the allocator after the runtime has shut down.  One entry per
    //  device, i.e. the last list and its index -- 36 bytes per slot -- stay allocated until the next differentiable getNeighborPairs call.)
    static std::mutex guard;
    static std::vector<PairIndexCache>* slots = new std::vector<PairIndexCache>(64);
    std::lock_guard<std::mutex> lock(guard);
    return (*slots)[(size_t)std::max(0, std::min(device, 63))];
}

class NeighborPairsFunction : public torch::autograd::Function<NeighborPairsFunction> {
public:
    static tensor_list forward(AutogradContext* ctx, const Tensor& positions, const torch::Scalar& cutoff,
                               const torch::Scalar& max_num_pairs, const Tensor& box_vectors, bool checkErrors) {
        // checks and messages of the reference (getNeighborPairsCUDA.cu:112-126,145-146)
        TORCH_CHECK(positions.dim() == 2, "Expected \"positions\" to have two dimensions");
        TORCH_CHECK(positions.size(0) > 0, "Expected the 1nd dimension size of \"positions\" to be more than 0");
        TORCH_CHECK(positions.size(1) == 3, "Expected the 2nd dimension size of \"positions\" to be 3");
        TORCH_CHECK(positions.is_contiguous(), "Expected \"positions\" to be contiguous");
        TORCH_CHECK(positions.scalar_type() == torch::kFloat32 || positions.scalar_type() == torch::kFloat64,
                    "Expected \"positions\" to be float32 or float64");
        const int64_t max_pairs = max_num_pairs.toLong();
        TORCH_CHECK(max_pairs > 0 || max_pairs == -1, "Expected \"max_num_pairs\" to be positive or equal to -1");
        TORCH_CHECK(cutoff.toDouble() > 0, "Expected \"cutoff\" to be positive");
        const bool use_periodic = box_vectors.size(0) != 0;
        Tensor box;
        if (use_periodic) {
            TORCH_CHECK(box_vectors.dim() == 2, "Expected \"box_vectors\" to have two dimensions");
            TORCH_CHECK(box_vectors.size(0) == 3 && box_vectors.size(1) == 3, "Expected \"box_vectors\" to have shape (3, 3)");
            box = box_vectors.to(positions.options()).contiguous();
        }
        const int64_t num_atoms = positions.size(0);
        // (argument limits of the C ABI, checked before anything is allocated)
        TORCH_CHECK(num_atoms <= std::numeric_limits<int32_t>::max(), "Too many atoms for getNeighborPairs");
        TORCH_CHECK(max_pairs != -1 || num_atoms <= 65536,
                    "max_num_pairs == -1 needs one slot per pair; beyond 65536 atoms use a compacted list");
        const int64_t slots = max_pairs == -1 ? num_atoms * (num_atoms - 1) / 2 : max_pairs;
        const auto options = positions.options();
        Tensor neighbors = torch::empty({2, slots}, options.dtype(torch::kInt32));
        Tensor deltas = torch::empty({slots, 3}, options);
        Tensor distances = torch::empty({slots}, options);
        Tensor num_pairs = torch::empty({1}, options.dtype(torch::kInt32));
        Tensor workspace = torch::empty({nnpops_neighbor_pairs_workspace_bytes((int)num_atoms)}, options.dtype(torch::kUInt8));
        const int dtype = positions.scalar_type() == torch::kFloat64 ? 1 : 0;
        c10::hip::HIPGuard guard(positions.device().index());
        void* stream = current_stream(positions.device());
        if (nnpops_neighbor_pairs_forward(dtype, (int)num_atoms, positions.data_ptr(), use_periodic ? box.data_ptr() : nullptr,
                                          cutoff.toDouble(), max_pairs, neighbors.data_ptr<int32_t>(), deltas.data_ptr(),
                                          distances.data_ptr(), num_pairs.data_ptr<int32_t>(), workspace.data_ptr(), stream) != NNPOPS_OK)
            raise_last("neighbors::getNeighborPairs");
        if (checkErrors) {      // synchronises: incompatible with graph capture, as documented by the reference (:156-160)
            const int found = num_pairs.item<int32_t>();
            TORCH_CHECK(found <= slots, "Too many neighbor pairs found. Maximum is " + std::to_string(slots),
                        " but found " + std::to_string(found));
        }
        // Round 6: the backward pass of a COMPACTED list is an owner-computes gather over the list's transposed index (no atomics,
        // include/nnpops_hip.h: nnpops_neighbor_pairs_build_index).  The index is a function of `neighbors` alone and the list is
        // grouped by row as this op emits it, so it is built here, once, when a gradient can be asked for, and travels with the saved
        // tensors (which autograd protects against in-place edits).  max_num_pairs == -1 (one slot per candidate pair, small systems)
        // and $NNPOPS_PAIRS_BACKWARD=fixed keep the order-independent fixed-point sums, which assume nothing about the list.
        Tensor index;
        const char* bwd_env = std::getenv("NNPOPS_PAIRS_BACKWARD");
        const bool fixed_point_only = bwd_env && std::string(bwd_env) == "fixed";
        if (positions.requires_grad() && max_pairs > 0 && !fixed_point_only && num_atoms <= NNPOPS_PAIRS_INDEX_MAX_ATOMS) {
            index = torch::empty({nnpops_neighbor_pairs_index_ints((int)num_atoms, slots)}, options.dtype(torch::kInt32));
            Tensor iws = torch::empty({nnpops_neighbor_pairs_index_workspace_bytes((int)num_atoms, slots) / 8 + 1}, options.dtype(torch::kInt64));
            if (nnpops_neighbor_pairs_build_index((int)num_atoms, slots, neighbors.data_ptr<int32_t>(), index.data_ptr<int32_t>(), iws.data_ptr(),
                                                  stream) != NNPOPS_OK)
                raise_last("neighbors::getNeighborPairs (transposed index)");
            PairIndexCache& cache = pair_index_cache(positions.device().index());
            cache.neighbors = neighbors; cache.index = index; cache.version = neighbors._version();
        }
        ctx->save_for_backward({neighbors, deltas, distances, index});
        ctx->saved_data["num_atoms"] = num_atoms;
        return {neighbors, deltas, distances, num_pairs};
    }

    static tensor_list backward(AutogradContext* ctx, tensor_list grad_outputs) {
        const auto saved = ctx->get_saved_variables();
        const Tensor neighbors = saved[0], deltas = saved[1], distances = saved[2], index = saved[3];
        const int64_t num_atoms = ctx->saved_data["num_atoms"].toInt();
        const Tensor grad_deltas = grad_outputs[1].defined() ? grad_outputs[1].contiguous() : torch::zeros_like(deltas);
        const Tensor grad_distances = grad_outputs[2].defined() ? grad_outputs[2].contiguous() : torch::zeros_like(distances);
        Tensor grad_positions = torch::empty({num_atoms, 3}, deltas.options());
        const int dtype = deltas.scalar_type() == torch::kFloat64 ? 1 : 0;
        c10::hip::HIPGuard guard(deltas.device().index());
        if (index.defined()) {
            Tensor terms = torch::empty({nnpops_neighbor_pairs_backward_indexed_workspace_bytes(dtype, distances.size(0)) / 8 + 1},
                                        deltas.options().dtype(torch::kInt64));
            if (nnpops_neighbor_pairs_backward_indexed(dtype, (int)num_atoms, distances.size(0), neighbors.data_ptr<int32_t>(), deltas.data_ptr(),
                                                       distances.data_ptr(), grad_deltas.data_ptr(), grad_distances.data_ptr(),
                                                       index.data_ptr<int32_t>(), grad_positions.data_ptr(), terms.data_ptr(),
                                                       current_stream(deltas.device())) != NNPOPS_OK)
                raise_last("neighbors::getNeighborPairs backward");
            return {grad_positions, Tensor(), Tensor(), Tensor(), Tensor()};
        }
        // (scratch for the order-independent fixed-point sums of the backward pass: no float atomics, nnpops_hip.h)
        Tensor workspace = torch::empty({nnpops_neighbor_pairs_backward_workspace_bytes((int)num_atoms) / 8}, deltas.options().dtype(torch::kInt64));
        if (nnpops_neighbor_pairs_backward_ws(dtype, (int)num_atoms, distances.size(0), neighbors.data_ptr<int32_t>(), deltas.data_ptr(),
                                              distances.data_ptr(), grad_deltas.data_ptr(), grad_distances.data_ptr(),
                                              grad_positions.data_ptr(), workspace.data_ptr(), current_stream(deltas.device())) != NNPOPS_OK)
            raise_last("neighbors::getNeighborPairs backward");
        return {grad_positions, Tensor(), Tensor(), Tensor(), Tensor()};
    }
};

TORCH_LIBRARY(neighbors, m) {
    m.def("getNeighborPairs(Tensor positions, Scalar cutoff, Scalar max_num_neighbors, Tensor box_vectors, bool checkErrors) -> "
          "(Tensor neighbors, Tensor deltas, Tensor distances, Tensor num_pairs)");
}

std::tuple<Tensor, Tensor, Tensor, Tensor> neighbor_pairs_device_entry(const Tensor& positions, const torch::Scalar& cutoff,
                                                                       const torch::Scalar& max_num_pairs, const Tensor& box_vectors,
                                                                       bool checkErrors) {
    const tensor_list r = NeighborPairsFunction::apply(positions, cutoff, max_num_pairs, box_vectors, checkErrors);
    return std::make_tuple(r[0], r[1], r[2], r[3]);
}
TORCH_LIBRARY_IMPL(neighbors, AutogradCUDA, m) { m.impl("getNeighborPairs", neighbor_pairs_device_entry); }
// (the backend key itself: what runs below autograd -- torch.inference_mode(), AutoDispatchBelowAutograd)
TORCH_LIBRARY_IMPL(neighbors, CUDA, m) { m.impl("getNeighborPairs", neighbor_pairs_device_entry); }

// ---------------------------------------------------------------------------------------------
// Host tensors.  The reference registers a CPU kernel of this op next to the device one (reference
// src/pytorch/neighbors/getNeighborPairsCPU.cpp:19-108, a composition of differentiable ATen calls that materialises
// all N(N-1)/2 candidate pairs); a drop-in keeps that dispatch key alive.  This is NOT a fallback of the device path --
// device tensors never come here, and a missing HIP library still fails at import -- it is what `positions.cpu()`
// callers of the reference get.  Written as plain loops over the pairs (no N^2 temporaries), same results:
//   * pair k <-> (row, column < row) in the reference's tril order, delta = positions[row] - positions[column],
//     triclinic wrap z, y, x with one round() each (:66-68), distance in the positions' dtype;
//   * max_num_pairs == -1: every slot kept, pairs beyond the cutoff masked with -1 / NaN (:72-78);
//   * otherwise: pairs with distance <= cutoff in tril order, padded with -1 / NaN up to max_num_pairs and NOT
//     truncated beyond it; num_pairs reports the length after padding, as the reference's CPU kernel does (:97-98).
// ---------------------------------------------------------------------------------------------
template <typename T>
void neighbor_pairs_host(const Tensor& positions, const Tensor& box, double cutoff, int64_t max_pairs, bool check,
                         Tensor& neighbors, Tensor& deltas, Tensor& distances) {
    const int64_t n = positions.size(0);
    const T* pos = positions.data_ptr<T>();
    const bool periodic = box.defined();
    T b[3][3] = {{0, 0, 0}, {0, 0, 0}, {0, 0, 0}};
    if (periodic) {
        const Tensor bc = box.to(positions.scalar_type()).contiguous();
        for (int i = 0; i < 9; i++) b[i / 3][i % 3] = bc.data_ptr<T>()[i];
    }
    const T cut = (T)cutoff;
    auto pair = [&](int64_t row, int64_t col, T (&d)[3]) -> T {
        for (int c = 0; c < 3; c++) d[c] = pos[3 * row + c] - pos[3 * col + c];
        if (periodic)
            for (int axis = 2; axis >= 0; axis--) {
                const T s = std::round(d[axis] / b[axis][axis]);
                for (int c = 0; c < 3; c++) d[c] -= s * b[axis][c];
            }
        return std::sqrt(d[0] * d[0] + d[1] * d[1] + d[2] * d[2]);
    };
    const T nan = std::numeric_limits<T>::quiet_NaN();
    const auto iopt = positions.options().dtype(torch::kInt32);
    if (max_pairs == -1) {
        const int64_t slots = n * (n - 1) / 2;
        neighbors = torch::empty({2, slots}, iopt);
        deltas = torch::empty({slots, 3}, positions.options());
        distances = torch::empty({slots}, positions.options());
        int32_t* nb = neighbors.data_ptr<int32_t>();
        T* dl = deltas.data_ptr<T>();
        T* ds = distances.data_ptr<T>();
        int64_t k = 0;
        for (int64_t row = 1; row < n; row++)
            for (int64_t col = 0; col < row; col++, k++) {
                T d[3];
                const T r = pair(row, col, d);
                const bool keep = !(r > cut);                  // the reference masks `distances > cutoff`
                nb[k] = keep ? (int32_t)row : -1;
                nb[slots + k] = keep ? (int32_t)col : -1;
                for (int c = 0; c < 3; c++) dl[3 * k + c] = keep ? d[c] : nan;
                ds[k] = keep ? r : nan;
            }
        return;
    }
    int64_t found = 0;
    for (int64_t row = 1; row < n; row++)
        for (int64_t col = 0; col < row; col++) {
            T d[3];
            found += pair(row, col, d) <= cut ? 1 : 0;
        }
    if (check)
        TORCH_CHECK(found <= max_pairs, "The maximum number of pairs has been exceed! Increase \"max_num_pairs\"");
    const int64_t slots = std::max(found, max_pairs);
    neighbors = torch::full({2, slots}, -1, iopt);
    deltas = torch::full({slots, 3}, nan, positions.options());
    distances = torch::full({slots}, nan, positions.options());
    int32_t* nb = neighbors.data_ptr<int32_t>();
    T* dl = deltas.data_ptr<T>();
    T* ds = distances.data_ptr<T>();
    int64_t k = 0;
    for (int64_t row = 1; row < n; row++)
        for (int64_t col = 0; col < row; col++) {
            T d[3];
            const T r = pair(row, col, d);
            if (!(r <= cut)) continue;
            nb[k] = (int32_t)row;
            nb[slots + k] = (int32_t)col;
            for (int c = 0; c < 3; c++) dl[3 * k + c] = d[c];
            ds[k++] = r;
        }
}

template <typename T>
void neighbor_pairs_host_backward(const Tensor& neighbors, const Tensor& deltas, const Tensor& distances, const Tensor& gd,
                                  const Tensor& gr, Tensor& gpos) {
    const int64_t slots = distances.size(0);
    const int32_t* nb = neighbors.data_ptr<int32_t>();
    const T* dl = deltas.data_ptr<T>();
    const T* ds = distances.data_ptr<T>();
    const T* pgd = gd.data_ptr<T>();
    const T* pgr = gr.data_ptr<T>();
    T* out = gpos.data_ptr<T>();
    for (int64_t k = 0; k < slots; k++) {
        const int32_t row = nb[k], col = nb[slots + k];
        if (row < 0) continue;                                 // masked / padding slot: no gradient (as the device kernel)
        for (int c = 0; c < 3; c++) {
            const T g = pgd[3 * k + c] + (ds[k] > 0 ? dl[3 * k + c] / ds[k] * pgr[k] : (T)0);
            out[3 * row + c] += g;
            out[3 * col + c] -= g;
        }
    }
}

class NeighborPairsHostFunction : public torch::autograd::Function<NeighborPairsHostFunction> {
public:
    static tensor_list forward(AutogradContext* ctx, const Tensor& positions, const torch::Scalar& cutoff,
                               const torch::Scalar& max_num_pairs, const Tensor& box_vectors, bool checkErrors) {
        // checks and messages of the reference's CPU kernel (getNeighborPairsCPU.cpp:25-53)
        TORCH_CHECK(positions.dim() == 2, "Expected \"positions\" to have two dimensions");
        TORCH_CHECK(positions.size(0) > 0, "Expected the 1nd dimension size of \"positions\" to be more than 0");
        TORCH_CHECK(positions.size(1) == 3, "Expected the 2nd dimension size of \"positions\" to be 3");
        TORCH_CHECK(positions.is_contiguous(), "Expected \"positions\" to be contiguous");
        TORCH_CHECK(positions.scalar_type() == torch::kFloat32 || positions.scalar_type() == torch::kFloat64,
                    "Expected \"positions\" to be float32 or float64");
        const double c = cutoff.toDouble();
        TORCH_CHECK(c > 0, "Expected \"cutoff\" to be positive");
        Tensor box;
        if (box_vectors.size(0) != 0) {
            TORCH_CHECK(box_vectors.dim() == 2, "Expected \"box_vectors\" to have two dimensions");
            TORCH_CHECK(box_vectors.size(0) == 3 && box_vectors.size(1) == 3, "Expected \"box_vectors\" to have shape (3, 3)");
            const Tensor v64 = box_vectors.to(torch::kFloat64).contiguous();
            const double* v = v64.data_ptr<double>();
            TORCH_CHECK(v[1] == 0, "Invalid box vectors: box_vectors[0][1] != 0");
            TORCH_CHECK(v[2] == 0, "Invalid box vectors: box_vectors[0][2] != 0");
            TORCH_CHECK(v[5] == 0, "Invalid box vectors: box_vectors[1][2] != 0");
            TORCH_CHECK(v[0] >= 2 * c, "Invalid box vectors: box_vectors[0][0] < 2*cutoff");
            TORCH_CHECK(v[4] >= 2 * c, "Invalid box vectors: box_vectors[1][1] < 2*cutoff");
            TORCH_CHECK(v[8] >= 2 * c, "Invalid box vectors: box_vectors[2][2] < 2*cutoff");
            TORCH_CHECK(v[0] >= 2 * v[3], "Invalid box vectors: box_vectors[0][0] < 2*box_vectors[1][0]");
            TORCH_CHECK(v[0] >= 2 * v[6], "Invalid box vectors: box_vectors[0][0] < 2*box_vectors[2][0]");
            TORCH_CHECK(v[4] >= 2 * v[7], "Invalid box vectors: box_vectors[1][1] < 2*box_vectors[2][1]");
            box = box_vectors;
        }
        const int64_t max_pairs = max_num_pairs.toLong();
        TORCH_CHECK(max_pairs > 0 || max_pairs == -1, "Expected \"max_num_pairs\" to be positive or equal to -1");
        Tensor neighbors, deltas, distances;
        if (positions.scalar_type() == torch::kFloat64)
            neighbor_pairs_host<double>(positions, box, c, max_pairs, checkErrors, neighbors, deltas, distances);
        else
            neighbor_pairs_host<float>(positions, box, c, max_pairs, checkErrors, neighbors, deltas, distances);
        Tensor num_pairs = torch::empty({1}, positions.options().dtype(torch::kInt32));
        num_pairs.data_ptr<int32_t>()[0] = (int32_t)distances.size(0);
        ctx->save_for_backward({neighbors, deltas, distances});
        ctx->saved_data["num_atoms"] = positions.size(0);
        return {neighbors, deltas, distances, num_pairs};
    }

    static tensor_list backward(AutogradContext* ctx, tensor_list grad_outputs) {
        const auto saved = ctx->get_saved_variables();
        const Tensor neighbors = saved[0], deltas = saved[1], distances = saved[2];
        const int64_t num_atoms = ctx->saved_data["num_atoms"].toInt();
        const Tensor gd = grad_outputs[1].defined() ? grad_outputs[1].contiguous() : torch::zeros_like(deltas);
        const Tensor gr = grad_outputs[2].defined() ? grad_outputs[2].contiguous() : torch::zeros_like(distances);
        Tensor gpos = torch::zeros({num_atoms, 3}, deltas.options());
        if (deltas.scalar_type() == torch::kFloat64) neighbor_pairs_host_backward<double>(neighbors, deltas, distances, gd, gr, gpos);
        else neighbor_pairs_host_backward<float>(neighbors, deltas, distances, gd, gr, gpos);
        return {gpos, Tensor(), Tensor(), Tensor(), Tensor()};
    }
};

std::tuple<Tensor, Tensor, Tensor, Tensor> neighbor_pairs_host_entry(const Tensor& positions, const torch::Scalar& cutoff,
                                                                     const torch::Scalar& max_num_pairs, const Tensor& box_vectors,
                                                                     bool checkErrors) {
    const tensor_list r = NeighborPairsHostFunction::apply(positions, cutoff, max_num_pairs, box_vectors, checkErrors);
    return std::make_tuple(r[0], r[1], r[2], r[3]);
}
TORCH_LIBRARY_IMPL(neighbors, AutogradCPU, m) { m.impl("getNeighborPairs", neighbor_pairs_host_entry); }
// the reference registers under the CPU key (getNeighborPairsCPU.cpp:102-108): reachable below autograd too
TORCH_LIBRARY_IMPL(neighbors, CPU, m) { m.impl("getNeighborPairs", neighbor_pairs_host_entry); }

// =============================================================================================
// PME, direct-space part (reference src/pytorch/pme/pme.cpp:4, pmeCUDA.cu:30-100,236-290, pmeCPU.cpp:75-175): same op
// name and schema; the energy's autograd backward scales the derivatives computed in the forward pass, exactly as the
// reference does.  The reciprocal-space op (pme_reciprocal) is not built.
// =============================================================================================
class PmeDirectFunction : public torch::autograd::Function<PmeDirectFunction> {
public:
    static Tensor forward(AutogradContext* ctx, const Tensor& positions, const Tensor& charges, const Tensor& neighbors,
                          const Tensor& deltas, const Tensor& distances, const Tensor& exclusions, const torch::Scalar& alpha,
                          const torch::Scalar& coulomb) {
        TORCH_CHECK(positions.dim() == 2 && positions.size(1) == 3, "positions must have shape (atoms, 3)");
        TORCH_CHECK(charges.dim() == 1 && charges.size(0) == positions.size(0), "charges must be 1D, one per atom");
        TORCH_CHECK(neighbors.dim() == 2 && neighbors.size(0) == 2, "neighbors must have shape (2, pairs)");
        TORCH_CHECK(exclusions.dim() == 2 && exclusions.size(0) == positions.size(0), "exclusions must have shape (atoms, max_exclusions)");
        TORCH_CHECK(positions.scalar_type() == torch::kFloat32 && charges.scalar_type() == torch::kFloat32 &&
                    deltas.scalar_type() == torch::kFloat32 && distances.scalar_type() == torch::kFloat32, "pme_direct computes in float32");
        const int64_t n = positions.size(0), pairs = neighbors.size(1), max_excl = exclusions.size(1);
        TORCH_CHECK(deltas.dim() == 2 && deltas.size(0) == pairs && deltas.size(1) == 3, "deltas must have shape (pairs, 3)");
        TORCH_CHECK(distances.dim() == 1 && distances.size(0) == pairs, "distances must have shape (pairs)");
        TORCH_CHECK(at::isIntegralType(neighbors.scalar_type(), false) && at::isIntegralType(exclusions.scalar_type(), false),
                    "neighbors and exclusions must hold integer indices");
        for (const Tensor* t : {&charges, &neighbors, &deltas, &distances, &exclusions})
            TORCH_CHECK(t->device() == positions.device(), "pme_direct: every tensor must be on the device of positions (",
                        positions.device(), "), got ", t->device());
        const Tensor pos = positions.contiguous(), q = charges.contiguous(), nb = neighbors.to(torch::kInt32).contiguous(),
                     dl = deltas.contiguous(), ds = distances.contiguous(), ex = exclusions.to(torch::kInt32).contiguous();
        const auto opts = positions.options();
        Tensor energy = torch::empty({}, opts), pos_deriv = torch::empty({n, 3}, opts), charge_deriv = torch::empty({n}, opts);
        const float a = (float)alpha.toDouble(), k = (float)coulomb.toDouble();
        if (positions.is_cuda()) {
            c10::hip::HIPGuard guard(positions.device().index());
            const PairIndexCache& cache = pair_index_cache(positions.device().index());
            const bool indexed = cache.index.defined() && cache.neighbors.defined() && nb.data_ptr() == cache.neighbors.data_ptr() &&
                                 nb.sizes() == cache.neighbors.sizes() && nb._version() == cache.version &&
                                 cache.index.numel() == nnpops_neighbor_pairs_index_ints((int)n, pairs);
            if (indexed) {
                Tensor workspace = torch::empty({nnpops_pme_direct_indexed_workspace_bytes(pairs, (int)n)}, opts.dtype(torch::kUInt8));
                if (nnpops_pme_direct_indexed((int)n, pairs, (int)max_excl, pos.data_ptr<float>(), q.data_ptr<float>(), nb.data_ptr<int32_t>(),
                                              dl.data_ptr<float>(), ds.data_ptr<float>(), max_excl ? ex.data_ptr<int32_t>() : nullptr,
                                              cache.index.data_ptr<int32_t>(), a, k, energy.data_ptr<float>(), pos_deriv.data_ptr<float>(),
                                              charge_deriv.data_ptr<float>(), workspace.data_ptr(), current_stream(positions.device())) != NNPOPS_OK)
                    raise_last("pme::pme_direct");
            } else {
            Tensor workspace = torch::empty({nnpops_pme_direct_workspace_bytes(pairs, (int)n, (int)max_excl)}, opts.dtype(torch::kUInt8));
            if (nnpops_pme_direct((int)n, pairs, (int)max_excl, pos.data_ptr<float>(), q.data_ptr<float>(), nb.data_ptr<int32_t>(),
                                  dl.data_ptr<float>(), ds.data_ptr<float>(), max_excl ? ex.data_ptr<int32_t>() : nullptr, a, k,
                                  energy.data_ptr<float>(), pos_deriv.data_ptr<float>(), charge_deriv.data_ptr<float>(),
                                  workspace.data_ptr(), current_stream(positions.device())) != NNPOPS_OK)
                raise_last("pme::pme_direct");
            }
        } else {
            // host tensors: the reference registers a CPU kernel too (pmeCPU.cpp:75-163); plain loops, double energy
            pos_deriv.zero_();
            charge_deriv.zero_();
            const float* P = pos.data_ptr<float>(); const float* Q = q.data_ptr<float>();
            const int32_t* N0 = nb.data_ptr<int32_t>(); const int32_t* N1 = N0 + pairs; const int32_t* E = ex.data_ptr<int32_t>();
            float* PD = pos_deriv.data_ptr<float>(); float* CD = charge_deriv.data_ptr<float>();
            const float* DL = dl.data_ptr<float>(); const float* DS = ds.data_ptr<float>();
            const float two_over_sqrt_pi = 1.12837916709551257390f;
            double e = 0.0;
            for (int64_t i = 0; i < pairs; i++) {
                const int a1 = N0[i], a2 = N1[i];
                TORCH_CHECK(a1 < n && a2 < n && (a1 < 0 || a2 >= 0), "pme_direct: neighbor index out of range at pair ", i);
                bool include = a1 > -1;
                for (int64_t j = 0; include && j < max_excl && E[a1 * max_excl + j] >= a2; j++)
                    if (E[a1 * max_excl + j] == a2) include = false;
                if (!include) continue;
                const float r = DS[i], inv_r = 1 / r, ar = a * r, pre = k * inv_r, er = std::erfc(ar);
                e += pre * er * Q[a1] * Q[a2];
                CD[a1] += pre * er * Q[a2];
                CD[a2] += pre * er * Q[a1];
                const float dedr = pre * Q[a1] * Q[a2] * (er + ar * std::exp(-ar * ar) * two_over_sqrt_pi) * inv_r * inv_r;
                for (int c = 0; c < 3; c++) { PD[3 * a1 + c] -= dedr * DL[3 * i + c]; PD[3 * a2 + c] += dedr * DL[3 * i + c]; }
            }
            for (int64_t a1 = 0; a1 < n; a1++)
                for (int64_t j = 0; j < max_excl && E[a1 * max_excl + j] > a1; j++) {
                    const int a2 = E[a1 * max_excl + j];
                    TORCH_CHECK(a2 < n, "pme_direct: exclusion index out of range for atom ", a1);
                    float d[3];
                    for (int c = 0; c < 3; c++) d[c] = P[3 * a1 + c] - P[3 * a2 + c];
                    const float r = std::sqrt(d[0] * d[0] + d[1] * d[1] + d[2] * d[2]), inv_r = 1 / r, ar = a * r, pre = k * inv_r, er = std::erf(ar);
                    e -= pre * er * Q[a1] * Q[a2];
                    CD[a1] -= pre * er * Q[a2];
                    CD[a2] -= pre * er * Q[a1];
                    const float dedr = pre * Q[a1] * Q[a2] * (er - ar * std::exp(-ar * ar) * two_over_sqrt_pi) * inv_r * inv_r;
                    for (int c = 0; c < 3; c++) { PD[3 * a1 + c] += dedr * d[c]; PD[3 * a2 + c] -= dedr * d[c]; }
                }
            energy.fill_((float)e);
        }
        ctx->save_for_backward({pos_deriv, charge_deriv});
        return energy;
    }

    static tensor_list backward(AutogradContext* ctx, tensor_list grad_outputs) {
        // The derivatives were computed (and detached) in the forward pass: a backward pass that is itself being recorded
        // (create_graph=True: force matching, Hessians) would silently see d(force)/d(positions, charges) = 0 from here.
        TORCH_CHECK(!torch::GradMode::is_enabled(),
                    "pme_direct: second derivatives are not implemented (backward was called with create_graph=True)");
        const auto saved = ctx->get_saved_variables();
        return {saved[0] * grad_outputs[0], saved[1] * grad_outputs[0], Tensor(), Tensor(), Tensor(), Tensor(), Tensor(), Tensor()};
    }
};

TORCH_LIBRARY(pme, m) {
    m.def("pme_direct(Tensor positions, Tensor charges, Tensor neighbors, Tensor deltas, Tensor distances, Tensor exclusions, "
          "Scalar alpha, Scalar coulomb) -> Tensor");
}

Tensor pme_direct_entry(const Tensor& positions, const Tensor& charges, const Tensor& neighbors, const Tensor& deltas,
                        const Tensor& distances, const Tensor& exclusions, const torch::Scalar& alpha, const torch::Scalar& coulomb) {
    return PmeDirectFunction::apply(positions, charges, neighbors, deltas, distances, exclusions, alpha, coulomb);
}

TORCH_LIBRARY_IMPL(pme, AutogradCUDA, m) { m.impl("pme_direct", pme_direct_entry); }
TORCH_LIBRARY_IMPL(pme, AutogradCPU, m) { m.impl("pme_direct", pme_direct_entry); }
// ... and the backend keys themselves (the reference registers its autograd Function under CPU, pmeCPU.cpp:381): below
// autograd -- torch.inference_mode(), AutoDispatchBelowAutograd -- the same entry runs without recording a graph
TORCH_LIBRARY_IMPL(pme, CUDA, m) { m.impl("pme_direct", pme_direct_entry); }
TORCH_LIBRARY_IMPL(pme, CPU, m) { m.impl("pme_direct", pme_direct_entry); }

// =============================================================================================
// BatchedLinear (reference src/pytorch/BatchedNN.cpp:30-50): y = W v + b broadcast over
// [molecules, atoms, models]; the backward skips the parameter gradients.
// =============================================================================================
class BatchedLinearFunction : public torch::autograd::Function<BatchedLinearFunction> {
public:
    static Tensor forward(AutogradContext* ctx, const Tensor& vectors, const Tensor& weights, const Tensor& biases) {
        ctx->save_for_backward({weights});
        return torch::matmul(weights, vectors) + biases;
    }
    static tensor_list backward(AutogradContext* ctx, const tensor_list& grads) {
        const Tensor weights = ctx->get_saved_variables()[0];
        // dL/dv = W^T dL/dy, written as a row-vector product so no transpose is materialised
        const Tensor row = grads[0].squeeze(-1).unsqueeze(-2);
        return {torch::matmul(row, weights).squeeze(-2).unsqueeze(-1), Tensor(), Tensor()};
    }
};

Tensor BatchedLinear(const Tensor& vectors, const Tensor& weights, const Tensor& biases) {
    return BatchedLinearFunction::apply(vectors, weights, biases);
}

// =============================================================================================
// GroupedMLP: the atomic networks of one frame, atoms grouped by species, on the split-fp16 GEMM of the C ABI
// (nnpops_gemm_split, batched_nn.hip).  Same function as BatchedNN.py:100-122 of the reference -- Linear, CELU(0.1),
// Linear, CELU, Linear, CELU, Linear per atom and ensemble member -- with bias + CELU fused into the GEMM epilogues
// and CELU' into the epilogues / prologue of the input-gradient pass: six GEMM launches per species and step, no
// elementwise kernels in between.  Returns the per-atom energies (atoms' own order) summed over the ensemble members.
//   x          [atoms, F] fp32, atoms in their own order; order [atoms] int32: atoms grouped by kind (the layer-0 GEMM
//              reads its rows through it, the last backward GEMM writes through it)      group_sizes  atoms per kind
//   fwd_*      planes of the weights, per kind: [M*H1][Fp] | [M*H2][H1p] | [M*H3][H2p]   (p: rounded up to 32)
//   bwd_*      planes of their transposes, per kind: [F][(M*H1)p] | M x [H1][H2p] | M x [H2][H3p]
//   biases     per kind: [M*H1] | [M*H2] | [M*H3]           last_w per kind [M*H3], last_b per kind (summed over members)
// =============================================================================================
constexpr float kCeluAlpha = 0.1f;          // BatchedNN.py:103
constexpr float kOperandScale = 1.0f / 16;   // operands are split after this scale: |activation| up to 1e6 stays in fp16 range

inline int64_t up32(int64_t v) { return (v + 31) / 32 * 32; }

struct MlpLayout {
    int64_t F, M, H1, H2, H3;
    int64_t fwd_kind() const { return M * H1 * up32(F) + M * H2 * up32(H1) + M * H3 * up32(H2); }
    int64_t bwd_kind() const { return F * up32(M * H1) + M * H1 * up32(H2) + M * H2 * up32(H3); }
    int64_t bias_kind() const { return M * (H1 + H2 + H3); }
};

void gemm_checked(int rc) { TORCH_CHECK(rc == NNPOPS_OK, nnpops_last_error()); }

class GroupedMLPFunction : public torch::autograd::Function<GroupedMLPFunction> {
public:
    static Tensor forward(AutogradContext* ctx, const Tensor& x, const Tensor& order, std::vector<int64_t> group_sizes, int64_t num_models, int64_t h1,
                          int64_t h2, int64_t h3, const Tensor& fwd_hi, const Tensor& fwd_lo, const Tensor& bwd_hi,
                          const Tensor& bwd_lo, const Tensor& biases, const Tensor& last_w, std::vector<double> last_b_host) {
        require_device_tensor(x, "x");
        TORCH_CHECK(x.dim() == 2 && x.scalar_type() == torch::kFloat32 && x.is_contiguous(), "x must be a contiguous [atoms, features] float32 tensor");
        TORCH_CHECK(order.dim() == 1 && order.size(0) == x.size(0) && order.scalar_type() == torch::kInt32 && order.is_contiguous() && order.device() == x.device(),
                    "order must be an int32 permutation of the atoms on the same device");
        const MlpLayout L{x.size(1), num_models, h1, h2, h3};
        const int64_t kinds = (int64_t)group_sizes.size(), atoms = x.size(0);
        TORCH_CHECK(fwd_hi.numel() == kinds * L.fwd_kind() && bwd_hi.numel() == kinds * L.bwd_kind() &&
                    biases.numel() == kinds * L.bias_kind() && last_w.numel() == kinds * L.M * L.H3,
                    "GroupedMLP: packed parameter buffers do not match the layer widths");
        c10::hip::HIPGuard guard(x.device().index());
        void* stream = current_stream(x.device());
        const auto opts = x.options();
        Tensor y1 = torch::empty({atoms, L.M * L.H1}, opts), y2 = torch::empty({atoms, L.M * L.H2}, opts), y3 = torch::empty({atoms, L.M * L.H3}, opts);
        Tensor energies = torch::empty({atoms}, opts);
        const at::Half* fh = fwd_hi.data_ptr<at::Half>(); const at::Half* fl = fwd_lo.data_ptr<at::Half>();
        const float* bs = biases.data_ptr<float>();
        TORCH_CHECK((int64_t)last_b_host.size() == kinds, "GroupedMLP: one last-layer bias per kind");
        int64_t first = 0;
        for (int64_t k = 0; k < kinds; k++) {
            const int64_t n = group_sizes[k];
            if (n > 0) {
                const at::Half *h0 = fh + k * L.fwd_kind(), *l0 = fl + k * L.fwd_kind();
                const at::Half *h1p = h0 + L.M * L.H1 * up32(L.F), *l1p = l0 + L.M * L.H1 * up32(L.F);
                const at::Half *h2p = h1p + L.M * L.H2 * up32(L.H1), *l2p = l1p + L.M * L.H2 * up32(L.H1);
                const float* b0 = bs + k * L.bias_kind(); const float* b1 = b0 + L.M * L.H1; const float* b2 = b1 + L.M * L.H2;
                const int* rows = order.data_ptr<int>() + first;
                float* p1 = y1.data_ptr<float>() + first * L.M * L.H1;
                float* p2 = y2.data_ptr<float>() + first * L.M * L.H2;
                float* p3 = y3.data_ptr<float>() + first * L.M * L.H3;
                // layer 0: every member reads the same AEVs -> one GEMM, N = members * H1
                gemm_checked(nnpops_gemm_split(stream, n, L.M * L.H1, L.F, 1, x.data_ptr<float>(), L.F, 0, h0, l0, up32(L.F), 0, p1, L.M * L.H1, 0, 1, b0, 0,
                                               nullptr, 0, 0, 0, nullptr, 0, 0, nullptr, 0, kCeluAlpha, kOperandScale, rows, nullptr));
                // layers 2 and 4: one problem per member
                gemm_checked(nnpops_gemm_split(stream, n, L.H2, L.H1, L.M, p1, L.M * L.H1, L.H1, h1p, l1p, up32(L.H1), L.H2 * up32(L.H1), p2,
                                               L.M * L.H2, L.H2, 1, b1, L.H2, nullptr, 0, 0, 0, nullptr, 0, 0, nullptr, 0, kCeluAlpha, kOperandScale, nullptr, nullptr));
                gemm_checked(nnpops_gemm_split(stream, n, L.H3, L.H2, L.M, p2, L.M * L.H2, L.H2, h2p, l2p, up32(L.H2), L.H3 * up32(L.H2), p3,
                                               L.M * L.H3, L.H3, 1, b2, L.H3, nullptr, 0, 0, 0, nullptr, 0, 0, nullptr, 0, kCeluAlpha, kOperandScale, nullptr, nullptr));
                // layer 6: one output per member, summed over the members
                gemm_checked(nnpops_rows_dot(stream, n, L.M * L.H3, p3, L.M * L.H3, last_w.data_ptr<float>() + k * L.M * L.H3, last_b_host[k],
                                             energies.data_ptr<float>(), rows));
            }
            first += n;
        }
        TORCH_CHECK(first == atoms, "GroupedMLP: group sizes do not add up to the number of atoms");
        ctx->save_for_backward({y1, y2, y3, bwd_hi, bwd_lo, last_w, order});
        ctx->saved_data["group_sizes"] = group_sizes;
        ctx->saved_data["dims"] = std::vector<int64_t>{L.F, L.M, L.H1, L.H2, L.H3};
        return energies;
    }

    static tensor_list backward(AutogradContext* ctx, const tensor_list& grads) {
        const auto saved = ctx->get_saved_variables();
        const Tensor &y1 = saved[0], &y2 = saved[1], &y3 = saved[2], &bwd_hi = saved[3], &bwd_lo = saved[4], &last_w = saved[5], &order = saved[6];
        const std::vector<int64_t> group_sizes = ctx->saved_data["group_sizes"].toIntVector();
        const std::vector<int64_t> d = ctx->saved_data["dims"].toIntVector();
        const MlpLayout L{d[0], d[1], d[2], d[3], d[4]};
        const int64_t atoms = y1.size(0);
        c10::hip::HIPGuard guard(y1.device().index());
        void* stream = current_stream(y1.device());
        const auto opts = y1.options();
        Tensor d2 = torch::empty({atoms, L.M * L.H2}, opts), d1 = torch::empty({atoms, L.M * L.H1}, opts), dx = torch::empty({atoms, L.F}, opts);
        const at::Half* bh = bwd_hi.data_ptr<at::Half>(); const at::Half* bl = bwd_lo.data_ptr<at::Half>();
        int64_t first = 0;
        for (size_t k = 0; k < group_sizes.size(); k++) {
            const int64_t n = group_sizes[k];
            if (n > 0) {
                const at::Half *t0h = bh + k * L.bwd_kind(), *t0l = bl + k * L.bwd_kind();               // [F][(M*H1)p]
                const at::Half *t1h = t0h + L.F * up32(L.M * L.H1), *t1l = t0l + L.F * up32(L.M * L.H1);  // M x [H1][H2p]
                const at::Half *t2h = t1h + L.M * L.H1 * up32(L.H2), *t2l = t1l + L.M * L.H1 * up32(L.H2);  // M x [H2][H3p]
                const float* p1 = y1.data_ptr<float>() + first * L.M * L.H1;
                const float* p2 = y2.data_ptr<float>() + first * L.M * L.H2;
                const float* p3 = y3.data_ptr<float>() + first * L.M * L.H3;
                float* q2 = d2.data_ptr<float>() + first * L.M * L.H2;
                float* q1 = d1.data_ptr<float>() + first * L.M * L.H1;
                const float* w6 = last_w.data_ptr<float>() + k * L.M * L.H3;
                // dE/dy3 = w6 * CELU'(y3) is formed while it is staged (prologue); times W4, times CELU'(y2)
                gemm_checked(nnpops_gemm_split(stream, n, L.H2, L.H3, L.M, nullptr, 0, 0, t2h, t2l, up32(L.H3), L.H2 * up32(L.H3), q2, L.M * L.H2,
                                               L.H2, 2, nullptr, 0, p2, L.M * L.H2, L.H2, 1, p3, L.M * L.H3, L.H3, w6, L.H3, kCeluAlpha, kOperandScale, nullptr, nullptr));
                gemm_checked(nnpops_gemm_split(stream, n, L.H1, L.H2, L.M, q2, L.M * L.H2, L.H2, t1h, t1l, up32(L.H2), L.H1 * up32(L.H2), q1, L.M * L.H1,
                                               L.H1, 2, nullptr, 0, p1, L.M * L.H1, L.H1, 0, nullptr, 0, 0, nullptr, 0, kCeluAlpha, kOperandScale, nullptr, nullptr));
                // all members' first layers at once: K = members * H1
                gemm_checked(nnpops_gemm_split(stream, n, L.F, L.M * L.H1, 1, q1, L.M * L.H1, 0, t0h, t0l, up32(L.M * L.H1), 0,
                                               dx.data_ptr<float>(), L.F, 0, 0, nullptr, 0, nullptr, 0, 0, 0, nullptr, 0, 0, nullptr, 0,
                                               kCeluAlpha, kOperandScale, nullptr, order.data_ptr<int>() + first));
            }
            first += n;
        }
        Tensor gx = dx * grads[0].unsqueeze(1);             // upstream gradient of every atom's energy (atoms' own order)
        return {gx, Tensor(), Tensor(), Tensor(), Tensor(), Tensor(), Tensor(), Tensor(), Tensor(), Tensor(), Tensor(), Tensor(), Tensor(), Tensor()};
    }
};

Tensor GroupedMLP(const Tensor& x, const Tensor& order, std::vector<int64_t> group_sizes, int64_t num_models, int64_t h1, int64_t h2, int64_t h3,
                  const Tensor& fwd_hi, const Tensor& fwd_lo, const Tensor& bwd_hi, const Tensor& bwd_lo, const Tensor& biases,
                  const Tensor& last_w, std::vector<double> last_b) {
    return GroupedMLPFunction::apply(x, order, group_sizes, num_models, h1, h2, h3, fwd_hi, fwd_lo, bwd_hi, bwd_lo, biases, last_w, last_b);
}

// =============================================================================================
// FusedMLP: the atomic networks of one frame on mlp_fused.hip -- sum over atoms AND members of the networks' outputs
// (BatchedNN.py:100-111 up to the division by the number of members).  When x requires a gradient the input-gradient pass
// runs inside forward and backward is one multiplication (see EnergyFunction above).
// =============================================================================================
class FusedMLPFunction : public torch::autograd::Function<FusedMLPFunction> {
public:
    static Tensor forward(AutogradContext* ctx, const Tensor& x, const Tensor& rows, std::vector<int64_t> kind_atoms, std::vector<int64_t> widths,
                          int64_t members, const Tensor& planes, const Tensor& floats, bool need_gradient, int64_t act_scale_log2) {
        c10::hip::HIPGuard guard(x.device().index());
        void* stream = current_stream(x.device());
        MlpCall call = mlp_prepare(x, rows, kind_atoms, widths, members, planes, floats, need_gradient, c10::nullopt, c10::nullopt, act_scale_log2);
        if (nnpops_mlp_forward(stream, &call.frame, need_gradient ? 1 : 0) != NNPOPS_OK) raise_last("NNPOpsBatchedNN::FusedMLP");
        Tensor total = torch::empty({1}, x.options());
        if (nnpops_mlp_energy_mean(stream, call.energies.data_ptr<float>(), call.energies.numel(), 1.0f, total.data_ptr<float>()) != NNPOPS_OK)
            raise_last("NNPOpsBatchedNN::FusedMLP");
        if (need_gradient) {
            Tensor dx = torch::empty_like(x);
            call.frame.dx = dx.data_ptr<float>(); call.frame.lddx = (int)dx.size(1); call.frame.dx_scale = 1.0f;
            if (nnpops_mlp_input_grad(stream, &call.frame) != NNPOPS_OK) raise_last("NNPOpsBatchedNN::FusedMLP");
            ctx->save_for_backward({dx});
        }
        return total;
    }
    static tensor_list backward(AutogradContext* ctx, const tensor_list& grads) {
        TORCH_CHECK(!torch::GradMode::is_enabled(), "NNPOpsBatchedNN::FusedMLP: second derivatives are not implemented (create_graph=True); "
                                                    "use layout='grouped' for that");
        const auto saved = ctx->get_saved_variables();
        TORCH_CHECK(!saved.empty(), "FusedMLP was evaluated without a gradient request");
        return {saved[0] * grads[0], Tensor(), Tensor(), Tensor(), Tensor(), Tensor(), Tensor(), Tensor(), Tensor()};
    }
};

Tensor FusedMLP(const Tensor& x, const Tensor& rows, std::vector<int64_t> kind_atoms, std::vector<int64_t> widths, int64_t members,
                const Tensor& planes, const Tensor& floats, int64_t act_scale_log2) {
    const bool need = torch::GradMode::is_enabled() && x.requires_grad();
    return FusedMLPFunction::apply(x, rows, kind_atoms, widths, members, planes, floats, need, act_scale_log2);
}

TORCH_LIBRARY(NNPOpsBatchedNN, m) {
    m.def("BatchedLinear", BatchedLinear);
    m.def("FusedMLP(Tensor x, Tensor rows, int[] kind_atoms, int[] widths, int members, Tensor planes, Tensor floats, int act_scale_log2=4) -> Tensor", FusedMLP);
    m.def("GroupedMLP(Tensor x, Tensor order, int[] group_sizes, int num_models, int h1, int h2, int h3, Tensor fwd_hi, Tensor fwd_lo, "
          "Tensor bwd_hi, Tensor bwd_lo, Tensor biases, Tensor last_w, float[] last_b) -> Tensor", GroupedMLP);
}

}  // namespace
