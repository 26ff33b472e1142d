// model.h - VMAF model loading for the host layer: a protocol-0 pickle reader for the one dict shape
// the reference's model files have (model/vmaf_v0.6.1.pkl), and the libsvm text model reader.
// Replaces ptools LoadValFromFile(..., SERIALIZE_P0) (wrapper/src/vmaf.cpp:57-136) and
// svm_load_model (wrapper/src/svm.cpp:468-571).  Nothing from the file is ever executed.
#ifndef VMAF_AMD_MODEL_H_
#define VMAF_AMD_MODEL_H_

#include <map>
#include <memory>
#include <stdexcept>
#include <string>
#include <vector>

namespace vmaf_amd {

struct PVal;
using PValPtr = std::shared_ptr<PVal>;

struct PVal {
    enum Kind { NONE, NUM, STR, LIST, DICT } kind = NONE;
    double num = 0;
    std::string str;
    std::vector<PValPtr> list;
    std::vector<std::pair<std::string, PValPtr>> dict;   // insertion order
    const PValPtr *find(const std::string &k) const
    {
        for (auto &kv : dict) if (kv.first == k) return &kv.second;
        return nullptr;
    }
};

PValPtr load_pickle_p0(const std::string &path);          // throws std::runtime_error

struct ScoreTransform {
    bool present = false;
    bool has_p0 = false, has_p1 = false, has_p2 = false;
    double p0 = 0, p1 = 0, p2 = 0;
    bool out_lte_in = false, out_gte_in = false;
};

struct VmafModel {
    std::vector<std::string> feature_names;               // e.g. "VMAF_feature_adm2_score"
    bool linear_rescale = false;
    std::vector<double> slopes, intercepts;               // index 0 = score, 1.. = features
    bool has_clip = false;
    double clip_lo = 0, clip_hi = 0;
    ScoreTransform transform;
    // libsvm nu-SVR, RBF
    double gamma = 0, rho = 0;
    int n_sv = 0, dim = 0;
    std::vector<double> sv;                               // dense [n_sv][dim]
    std::vector<double> coef;
};

// Loads <path> (.pkl) and <path>.model.  Throws std::runtime_error with the reference's messages.
VmafModel load_vmaf_model(const std::string &pkl_path);

}  // namespace vmaf_amd
#endif
