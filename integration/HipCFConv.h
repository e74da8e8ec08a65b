// HipCFConv.h -- the SchNet counterpart of HipANISymmetryFunctions.h: subclasses of the reference's abstract
// CFConvNeighbors / CFConv (src/schnet/CFConv.h:37-217) over the C ABI of libnnpops_hip.so.
#pragma once

#include "CFConv.h"                      // the reference's header
#include "HipStaging.h"

class HipCFConvNeighbors : public CFConvNeighbors {
public:
    HipCFConvNeighbors(int numAtoms, float cutoff, bool periodic, int device = 0) : CFConvNeighbors(numAtoms, cutoff, periodic) {
        nnpops_integration::abiTry(nnpops_cfconv_neighbors_create(&handle, numAtoms, cutoff, periodic, device));
    }
    ~HipCFConvNeighbors() override { nnpops_cfconv_neighbors_destroy(handle); }

    void build(const float* positions, const float* periodicBoxVectors) override {
        using namespace nnpops_integration;
        const float* dPos = pos.in(positions, 3 * (size_t)getNumAtoms());
        const float* dBox = getPeriodic() ? box.in(periodicBoxVectors, 9) : nullptr;
        if (getPeriodic() && periodicBoxVectors) {
            float h[9];
            hipTry(hipMemcpy(h, dBox, sizeof(h), hipMemcpyDefault), "box");
            triclinic = h[1] != 0 || h[2] != 0 || h[3] != 0 || h[5] != 0 || h[6] != 0 || h[7] != 0;   // CpuCFConv.cpp:72-77
        }
        for (int attempt = 0;; attempt++) {                  // a capacity error grows the buffers: build again (bounded)
            abiTry(nnpops_cfconv_neighbors_build(handle, dPos, dBox));
            const int rc = nnpops_cfconv_neighbors_check(handle, nullptr);
            if (rc == NNPOPS_OK) break;
            if (rc != NNPOPS_ERR_CAPACITY || attempt >= 8) abiTry(rc);
        }
    }
    bool getTriclinic() const override { return triclinic; }
    nnpops_cfconv_neighbors_t getHandle() const { return handle; }

private:
    nnpops_cfconv_neighbors_t handle = nullptr;
    bool triclinic = false;
    nnpops_integration::DeviceMirror pos, box;
};

class HipCFConv : public CFConv {
public:
    // w1 [width][numGaussians], w2 [width][width], b1/b2 [width]: host arrays, as for CpuCFConv (CFConv.h:125-138)
    HipCFConv(int numAtoms, int width, int numGaussians, float cutoff, bool periodic, float gaussianWidth,
              ActivationFunction activation, const float* w1, const float* b1, const float* w2, const float* b2, int device = 0)
        : CFConv(numAtoms, width, numGaussians, cutoff, periodic, gaussianWidth, activation) {
        nnpops_integration::abiTry(nnpops_cfconv_create(&handle, numAtoms, width, numGaussians, cutoff, periodic, gaussianWidth,
                                                       (int)activation, w1, b1, w2, b2, device));
    }
    ~HipCFConv() override { nnpops_cfconv_destroy(handle); }

    void compute(const CFConvNeighbors& neighbors, const float* positions, const float* periodicBoxVectors, const float* input,
                 float* output) override {
        using namespace nnpops_integration;
        const size_t n = (size_t)getNumAtoms(), w = (size_t)getWidth();
        const float* dPos = pos.in(positions, 3 * n);
        const float* dBox = getPeriodic() ? box.in(periodicBoxVectors, 9) : nullptr;
        const float* dIn = in.in(input, n * w);
        float* dOut = out.out(output, n * w);
        abiTry(nnpops_cfconv_compute(handle, hip(neighbors), dPos, dBox, dIn, dOut));
        hipTry(hipDeviceSynchronize(), "compute");
        out.finish();
    }

    void backprop(const CFConvNeighbors& neighbors, const float* positions, const float* periodicBoxVectors, const float* input,
                  const float* outputDeriv, float* inputDeriv, float* positionDeriv) override {
        using namespace nnpops_integration;
        const size_t n = (size_t)getNumAtoms(), w = (size_t)getWidth();
        const float* dPos = pos.in(positions, 3 * n);
        const float* dBox = getPeriodic() ? box.in(periodicBoxVectors, 9) : nullptr;
        const float* dIn = in.in(input, n * w);
        const float* dOutDeriv = outDeriv.in(outputDeriv, n * w);
        float* dInDeriv = inDeriv.out(inputDeriv, n * w);
        float* dPosDeriv = posDeriv.out(positionDeriv, 3 * n);
        abiTry(nnpops_cfconv_backprop(handle, hip(neighbors), dPos, dBox, dIn, dOutDeriv, dInDeriv, dPosDeriv));
        hipTry(hipDeviceSynchronize(), "backprop");
        inDeriv.finish();
        posDeriv.finish();
    }

private:
    static nnpops_cfconv_neighbors_t hip(const CFConvNeighbors& neighbors) {
        return dynamic_cast<const HipCFConvNeighbors&>(neighbors).getHandle();      // CudaCFConv does the same cast
    }
    nnpops_cfconv_t handle = nullptr;
    nnpops_integration::DeviceMirror pos, box, in, out, outDeriv, inDeriv, posDeriv;
};
