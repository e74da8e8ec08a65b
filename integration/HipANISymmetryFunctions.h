// HipANISymmetryFunctions.h -- the class a maintainer adds to the reference tree (next to
// src/ani/CudaANISymmetryFunctions.h) to put the MI355X path behind the reference's own core API.
// It subclasses the reference's abstract ANISymmetryFunctions (src/ani/ANISymmetryFunctions.h:41-154) and
// forwards to the C ABI of libnnpops_hip.so (include/nnpops_hip.h).  No device code here.
//
// Built against the reference headers in place (never copied): oracle/Makefile, target `ref_tests`, compiles
// the reference's own test suite (src/ani/TestANISymmetryFunctions.h) with this class as the implementation.
#pragma once

#include <vector>

#include "ANISymmetryFunctions.h"        // the reference's header
#include "HipStaging.h"

class HipANISymmetryFunctions : public ANISymmetryFunctions {
public:
    HipANISymmetryFunctions(int numAtoms, int numSpecies, float radialCutoff, float angularCutoff, bool periodic,
                            const std::vector<int>& atomSpecies, const std::vector<RadialFunction>& radialFunctions,
                            const std::vector<AngularFunction>& angularFunctions, bool torchani, int device = 0)
        : ANISymmetryFunctions(numAtoms, numSpecies, radialCutoff, angularCutoff, periodic, atomSpecies, radialFunctions,
                               angularFunctions, torchani) {
        static_assert(sizeof(RadialFunction) == 2 * sizeof(float) && sizeof(AngularFunction) == 4 * sizeof(float),
                      "parameter records are passed to the C ABI as packed floats");
        nnpops_integration::abiTry(nnpops_ani_create(
            &handle, numAtoms, numSpecies, radialCutoff, angularCutoff, periodic, atomSpecies.data(), (int)radialFunctions.size(),
            reinterpret_cast<const float*>(radialFunctions.data()), (int)angularFunctions.size(),
            reinterpret_cast<const float*>(angularFunctions.data()), torchani, device));
        widthRadial = (size_t)numSpecies * radialFunctions.size();
        widthAngular = (size_t)numSpecies * (numSpecies + 1) / 2 * angularFunctions.size();
    }
    ~HipANISymmetryFunctions() override { nnpops_ani_destroy(handle); }

    void setStream(void* stream) { nnpops_integration::abiTry(nnpops_ani_set_stream(handle, stream)); }   // cf. the CUDA class

    void computeSymmetryFunctions(const float* positions, const float* periodicBoxVectors, float* radial, float* angular) override {
        using namespace nnpops_integration;
        const size_t n = (size_t)getNumAtoms();
        const float* dPos = pos.in(positions, 3 * n);
        const float* dBox = getPeriodic() ? box.in(periodicBoxVectors, 9) : nullptr;
        float* dRad = rad.out(radial, n * widthRadial);
        float* dAng = ang.out(angular, n * widthAngular);
        for (;;) {          // neighbour buffers have a capacity; check() grows them and asks for a repeat
            abiTry(nnpops_ani_compute(handle, dPos, dBox, dRad, dAng));
            const int rc = nnpops_ani_check(handle, nullptr, nullptr);
            if (rc == NNPOPS_OK) break;
            if (rc != NNPOPS_ERR_CAPACITY) abiTry(rc);
        }
        rad.finish();
        ang.finish();
    }

    void backprop(const float* radialDeriv, const float* angularDeriv, float* positionDeriv) override {
        using namespace nnpops_integration;
        const size_t n = (size_t)getNumAtoms();
        const float* dRad = gradRad.in(radialDeriv, n * widthRadial);
        const float* dAng = gradAng.in(angularDeriv, n * widthAngular);
        float* dPos = gradPos.out(positionDeriv, 3 * n);
        abiTry(nnpops_ani_backprop(handle, dRad, dAng, dPos));
        hipTry(hipDeviceSynchronize(), "backprop");
        gradPos.finish();
    }

private:
    nnpops_ani_t handle = nullptr;
    size_t widthRadial = 0, widthAngular = 0;
    nnpops_integration::DeviceMirror pos, box, rad, ang, gradRad, gradAng, gradPos;
};
