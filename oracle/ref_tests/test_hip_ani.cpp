// The reference's own C++ test suite for the ANI symmetry functions (src/ani/TestANISymmetryFunctions.h: the
// TorchANI-generated 18-atom water goldens, periodic / triclinic variants, finite-difference derivative checks),
// instantiated with the MI355X implementation exactly as src/ani/TestCudaANISymmetryFunctions.cu instantiates it
// with the CUDA one.  Test infrastructure: built by `make -C oracle ref_tests` from the reference sources in
// place into oracle/_ref/, run by tests/test_reference_cpp_suites_gpu.py.
#include "HipANISymmetryFunctions.h"

ANISymmetryFunctions* createSymmetryCalculator(int numAtoms, int numSpecies, float radialCutoff, float angularCutoff, bool periodic,
                                               const std::vector<int>& atomSpecies, const std::vector<RadialFunction>& radialFunctions,
                                               const std::vector<AngularFunction>& angularFunctions, bool torchani) {
    return new HipANISymmetryFunctions(numAtoms, numSpecies, radialCutoff, angularCutoff, periodic, atomSpecies, radialFunctions,
                                       angularFunctions, torchani);
}

#include "TestANISymmetryFunctions.h"    // the reference's tests and main()
