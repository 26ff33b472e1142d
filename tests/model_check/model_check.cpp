// model_check.cpp - test driver for vmaf_amd/csrc/model.cpp on the CPU (no HIP): loads the model files named on the
// command line and prints one line per file, "ok <n_features> <n_sv> <dim>" or "error <message>".  Built with
// -fsanitize=address,undefined by tests/test_model_files.py (make -C vmaf_amd/csrc model_check_asan), so that a garbled
// or truncated .pkl / .model that makes the parsers read or write out of bounds fails the test instead of passing by luck.
// What the reference does with such files: throws "Error loading model (.pkl)" / "Error loading SVM model"
// (wrapper/src/vmaf.cpp:57-136, wrapper/src/svm.cpp:468-571) - an exception, never a crash.
#include <cstdio>
#include <exception>

#include "../../vmaf_amd/csrc/model.h"

int main(int argc, char **argv)
{
    for (int i = 1; i < argc; ++i) {
        try {
            const vmaf_amd::VmafModel m = vmaf_amd::load_vmaf_model(argv[i]);
            std::printf("ok %zu %d %d\n", m.feature_names.size(), m.n_sv, m.dim);
        } catch (const std::exception &e) {
            std::printf("error %s\n", e.what());
        }
        std::fflush(stdout);
    }
    return 0;
}
