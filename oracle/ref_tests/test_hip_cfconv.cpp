// The reference's own C++ test suite for CFConv (src/schnet/TestCFConv.h: SchNetPack-generated goldens,
// periodic / triclinic / tanh variants, derivative checks) run against the MI355X implementation, cf.
// src/schnet/TestCudaCFConv.cu.  Test infrastructure, see test_hip_ani.cpp.
#include "HipCFConv.h"

CFConvNeighbors* createNeighbors(int numAtoms, float cutoff, bool periodic) {
    return new HipCFConvNeighbors(numAtoms, cutoff, periodic);
}

CFConv* createConv(int numAtoms, int width, int numGaussians, float cutoff, bool periodic, float gaussianWidth,
                   CFConv::ActivationFunction activation, float* w1, float* b1, float* w2, float* b2) {
    return new HipCFConv(numAtoms, width, numGaussians, cutoff, periodic, gaussianWidth, activation, w1, b1, w2, b2);
}

#include "TestCFConv.h"                  // the reference's tests and main()
