/*
 * oracle/ref_shim.cpp -- TEST INFRASTRUCTURE ONLY.
 *
 * extern "C" doorway onto the REFERENCE's own CPU classes, compiled from the sources where
 * they lie under /root/reference (never copied into this repository).  oracle/Makefile
 * builds this file together with
 *     /root/reference/src/ani/CpuANISymmetryFunctions.cpp
 *     /root/reference/src/schnet/CpuCFConv.cpp
 * into oracle/_ref/libnnpops_ref.so.  The entry points mirror oracle/ani_oracle.c and
 * oracle/cfconv_oracle.c one-for-one (prefix ref_ instead of ani_oracle_/cfconv_oracle_) so
 * that tests can pin the restatement against the real thing, and bench.py can time the
 * reference CPU path itself (cpu_baseline.kind == "reference").
 */
#include "CpuANISymmetryFunctions.h"   // /root/reference/src/ani
#include "CpuCFConv.h"                 // /root/reference/src/schnet
#include <cstddef>
#include <vector>

extern "C" {

void* ref_ani_create(int n_atoms, int n_species, float rc_radial, float rc_angular, int periodic, const int* species,
                     int n_radial, const float* radial_eta_rs, int n_angular, const float* angular_eta_rs_zeta_ths,
                     int torchani) {
    std::vector<int> sp(species, species + n_atoms);
    std::vector<RadialFunction> rf;
    for (int k = 0; k < n_radial; k++) rf.push_back({radial_eta_rs[2 * k], radial_eta_rs[2 * k + 1]});
    std::vector<AngularFunction> af;
    for (int m = 0; m < n_angular; m++)
        af.push_back({angular_eta_rs_zeta_ths[4 * m], angular_eta_rs_zeta_ths[4 * m + 1],
                      angular_eta_rs_zeta_ths[4 * m + 2], angular_eta_rs_zeta_ths[4 * m + 3]});
    return new CpuANISymmetryFunctions(n_atoms, n_species, rc_radial, rc_angular, periodic != 0, sp, rf, af, torchani != 0);
}
void ref_ani_destroy(void* h) { delete static_cast<CpuANISymmetryFunctions*>(h); }
void ref_ani_forward(void* h, const float* pos, const float* box, float* radial, float* angular) {
    static_cast<CpuANISymmetryFunctions*>(h)->computeSymmetryFunctions(pos, box, radial, angular);
}
void ref_ani_backward(void* h, const float* radial_grad, const float* angular_grad, float* pos_grad) {
    static_cast<CpuANISymmetryFunctions*>(h)->backprop(radial_grad, angular_grad, pos_grad);
}

void* ref_cfconv_neighbors_create(int n_atoms, float cutoff, int periodic) {
    return new CpuCFConvNeighbors(n_atoms, cutoff, periodic != 0);
}
void ref_cfconv_neighbors_destroy(void* h) { delete static_cast<CpuCFConvNeighbors*>(h); }
void ref_cfconv_neighbors_build(void* h, const float* pos, const float* box) {
    static_cast<CpuCFConvNeighbors*>(h)->build(pos, box);
}
int ref_cfconv_neighbors_num_pairs(void* h) {
    int n = 0;
    for (const auto& row : static_cast<CpuCFConvNeighbors*>(h)->getNeighbors()) n += (int)row.size();
    return n;
}
/* flatten the half list into caller-provided arrays: start[n_atoms+1], other[P], dist[P] */
void ref_cfconv_neighbors_export(void* h, int* start, int* other, float* dist) {
    auto* nb = static_cast<CpuCFConvNeighbors*>(h);
    int p = 0;
    for (int i = 0; i < nb->getNumAtoms(); i++) {
        start[i] = p;
        const auto& row = nb->getNeighbors()[i];
        const auto& d = nb->getNeighborDistances()[i];
        for (size_t q = 0; q < row.size(); q++) { other[p] = row[q]; dist[p] = d[q]; p++; }
    }
    start[nb->getNumAtoms()] = p;
}

void* ref_cfconv_create(int n_atoms, int width, int n_gauss, float cutoff, int periodic, float sigma, int activation,
                        const float* w1, const float* b1, const float* w2, const float* b2) {
    return new CpuCFConv(n_atoms, width, n_gauss, cutoff, periodic != 0, sigma,
                         activation == 0 ? CFConv::ShiftedSoftplus : CFConv::Tanh, w1, b1, w2, b2);
}
void ref_cfconv_destroy(void* h) { delete static_cast<CpuCFConv*>(h); }
void ref_cfconv_forward(void* h, void* nb, const float* pos, const float* box, const float* input, float* output) {
    static_cast<CpuCFConv*>(h)->compute(*static_cast<CpuCFConvNeighbors*>(nb), pos, box, input, output);
}
void ref_cfconv_backward(void* h, void* nb, const float* pos, const float* box, const float* input,
                         const float* output_grad, float* input_grad, float* pos_grad) {
    static_cast<CpuCFConv*>(h)->backprop(*static_cast<CpuCFConvNeighbors*>(nb), pos, box, input, output_grad,
                                         input_grad, pos_grad);
}

}  // extern "C"
