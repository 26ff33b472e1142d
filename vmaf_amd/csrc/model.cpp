// model.cpp - see model.h
#include "model.h"

#include <algorithm>
#include <clocale>
#include <cmath>
#include <locale>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <sstream>

namespace vmaf_amd {

namespace {

std::string unquote(const std::string &t)
{
    size_t a = 0, b = t.size();
    while (b > a && (t[b - 1] == '\r' || t[b - 1] == ' ')) --b;
    if (b - a >= 2 && (t[a] == '\'' || t[a] == '"') && t[b - 1] == t[a]) return t.substr(a + 1, b - a - 2);
    return t.substr(a, b - a);
}

// first stack slot an opcode may touch: nothing below the innermost open mark
size_t floor_of(const std::vector<size_t> &marks) { return marks.empty() ? 0 : marks.back(); }

double parse_double_c(const std::string &t)
{
    // strtod under the "C" numeric convention regardless of the caller's locale
    std::istringstream is(t);
    is.imbue(std::locale::classic());
    double v = 0;
    is >> v;
    if (is.fail()) {
        if (t.find("inf") != std::string::npos) return t[0] == '-' ? -HUGE_VAL : HUGE_VAL;
        throw std::runtime_error("bad float in pickle: " + t);
    }
    return v;
}

}  // namespace

// Opcodes used by the reference's protocol-0 model files (model/vmaf_v0.6.1.pkl:1-108):
//   (  MARK   d DICT   l LIST   p PUT   g GET   S/V STRING   F FLOAT   I INT   N NONE
//   s  SETITEM   a APPEND   . STOP.  Anything else (notably GLOBAL/REDUCE) is refused.
PValPtr load_pickle_p0(const std::string &path)
{
    std::ifstream f(path);
    if (!f) throw std::runtime_error("cannot open " + path);
    std::vector<PValPtr> stack;
    std::vector<size_t> marks;
    std::map<int, PValPtr> memo;
    std::string line;
    while (std::getline(f, line)) {
        size_t pos = 0;
        while (pos < line.size()) {
            const char op = line[pos++];
            const std::string rest = line.substr(pos);
            switch (op) {
            case '(': marks.push_back(stack.size()); break;
            case 'd': case 'l': {
                if (marks.empty()) throw std::runtime_error("pickle: no mark");
                const size_t m = marks.back();
                marks.pop_back();
                if (m > stack.size()) throw std::runtime_error("pickle: mark above the stack");
                auto v = std::make_shared<PVal>();
                if (op == 'd') {
                    v->kind = PVal::DICT;
                    if ((stack.size() - m) % 2) throw std::runtime_error("pickle: odd number of dict items");
                    for (size_t i = m; i + 1 < stack.size(); i += 2) {
                        if (stack[i]->kind != PVal::STR) throw std::runtime_error("pickle: dict key is not a string");
                        v->dict.emplace_back(stack[i]->str, stack[i + 1]);
                    }
                } else {
                    v->kind = PVal::LIST;
                    v->list.assign(stack.begin() + m, stack.end());
                }
                stack.resize(m);
                stack.push_back(v);
                break;
            }
            case 'p': if (stack.empty() || stack.size() - 1 < floor_of(marks)) throw std::runtime_error("pickle: put on empty stack");
                      memo[std::atoi(rest.c_str())] = stack.back(); pos = line.size(); break;
            case 'g': { auto it = memo.find(std::atoi(rest.c_str()));
                        if (it == memo.end()) throw std::runtime_error("pickle: bad memo key");
                        stack.push_back(it->second); pos = line.size(); break; }
            case 'S': case 'V': { auto v = std::make_shared<PVal>(); v->kind = PVal::STR; v->str = unquote(rest);
                                  stack.push_back(v); pos = line.size(); break; }
            case 'F': { auto v = std::make_shared<PVal>(); v->kind = PVal::NUM; v->num = parse_double_c(rest);
                        stack.push_back(v); pos = line.size(); break; }
            case 'I': { auto v = std::make_shared<PVal>(); v->kind = PVal::NUM; v->num = (double)std::atol(rest.c_str());
                        stack.push_back(v); pos = line.size(); break; }
            case 'N': stack.push_back(std::make_shared<PVal>()); break;
            case 's': {
                // the dict, the key and the value all lie above the innermost open mark
                if (stack.size() < 3 || stack.size() - 3 < floor_of(marks)) throw std::runtime_error("pickle: setitem underflow");
                PValPtr val = stack.back(); stack.pop_back();
                PValPtr key = stack.back(); stack.pop_back();
                PVal &d = *stack.back();
                if (d.kind != PVal::DICT) throw std::runtime_error("pickle: setitem on non-dict");
                if (key->kind != PVal::STR) throw std::runtime_error("pickle: dict key is not a string");
                bool found = false;
                for (auto &kv : d.dict) if (kv.first == key->str) { kv.second = val; found = true; }
                if (!found) d.dict.emplace_back(key->str, val);
                break;
            }
            case 'a': {
                if (stack.size() < 2 || stack.size() - 2 < floor_of(marks)) throw std::runtime_error("pickle: append underflow");
                PValPtr val = stack.back(); stack.pop_back();
                if (stack.back()->kind != PVal::LIST) throw std::runtime_error("pickle: append on non-list");
                stack.back()->list.push_back(val);
                break;
            }
            case '.': if (stack.empty() || !marks.empty()) throw std::runtime_error("pickle: empty or open mark at stop");
                      return stack.back();
            case '\r': case ' ': break;
            default: throw std::runtime_error(std::string("pickle: unsupported opcode '") + op + "'");
            }
        }
    }
    throw std::runtime_error("pickle: no STOP opcode");
}

namespace {

struct Incompatible : std::runtime_error { using std::runtime_error::runtime_error; };

void load_svm_text(const std::string &path, VmafModel &m)
{
    FILE *fp = std::fopen(path.c_str(), "rb");
    if (!fp) throw std::runtime_error("Error loading SVM model");
    std::string text;
    char buf[65536];
    size_t n;
    while ((n = std::fread(buf, 1, sizeof(buf), fp)) > 0) text.append(buf, n);
    std::fclose(fp);

    std::istringstream is(text);
    is.imbue(std::locale::classic());
    std::string line;
    bool in_sv = false, nu_svr = false, rbf = false;
    int total_sv = -1;
    struct Row { double coef; std::vector<std::pair<int, double>> items; };
    std::vector<Row> rows;
    while (std::getline(is, line)) {
        std::istringstream ls(line);
        ls.imbue(std::locale::classic());
        if (!in_sv) {
            std::string key;
            if (!(ls >> key)) continue;
            if (key == "svm_type") { std::string v; ls >> v; nu_svr = (v == "nu_svr"); }
            else if (key == "kernel_type") { std::string v; ls >> v; rbf = (v == "rbf"); }
            else if (key == "gamma") ls >> m.gamma;
            else if (key == "rho") ls >> m.rho;
            else if (key == "total_sv") ls >> total_sv;
            else if (key == "SV") in_sv = true;
            else if (key == "degree" || key == "coef0" || key == "nr_class" || key == "label" || key == "probA" ||
                     key == "probB" || key == "nr_sv") { /* not used by nu-SVR RBF prediction */ }
            else throw std::runtime_error("Error loading SVM model");      // svm.cpp:457-461
        } else {
            Row r;
            if (!(ls >> r.coef)) continue;
            std::string tok;
            while (ls >> tok) {
                const size_t c = tok.find(':');
                if (c == std::string::npos) throw std::runtime_error("Error loading SVM model");
                r.items.emplace_back(std::atoi(tok.substr(0, c).c_str()), parse_double_c(tok.substr(c + 1)));
            }
            rows.push_back(std::move(r));
        }
    }
    if (!nu_svr || !rbf || total_sv <= 0 || (int)rows.size() < total_sv) throw std::runtime_error("Error loading SVM model");
    rows.resize(total_sv);
    int dim = 0;
    constexpr int kMaxIndex = 4096;                       // a garbled index must not size the dense table
    for (auto &r : rows) for (auto &it : r.items) { if (it.first < 1 || it.first > kMaxIndex) throw std::runtime_error("Error loading SVM model"); dim = std::max(dim, it.first); }
    if (dim == 0) throw std::runtime_error("Error loading SVM model");
    m.n_sv = total_sv;
    m.dim = dim;
    m.sv.assign((size_t)total_sv * dim, 0.0);             // omitted sparse entries are zero (svm.cpp:182-216)
    m.coef.resize(total_sv);
    for (int i = 0; i < total_sv; ++i) {
        m.coef[i] = rows[i].coef;
        for (auto &it : rows[i].items) m.sv[(size_t)i * dim + it.first - 1] = it.second;
    }
}

}  // namespace

VmafModel load_vmaf_model(const std::string &pkl_path)
{
    VmafModel m;
    PValPtr top;
    const PValPtr *md = nullptr;
    auto need = [&](const char *k) -> const PValPtr & {
        const PValPtr *p = (*md)->find(k);
        if (!p) throw std::runtime_error(std::string("Error loading model (.pkl): missing key ") + k);
        return *p;
    };
    try {
        top = load_pickle_p0(pkl_path);
        if (top->kind != PVal::DICT || !(md = top->find("model_dict")) || (*md)->kind != PVal::DICT)
            throw std::runtime_error("no model_dict");
    } catch (const std::runtime_error &e) {
        std::printf("Input model at %s cannot be read successfully.\n", pkl_path.c_str());   // vmaf.cpp:90-92
        throw std::runtime_error(std::string("Error loading model (.pkl): ") + e.what());
    }
    const PValPtr &model_type = need("model_type"), &feature_names = need("feature_names"), &norm_type = need("norm_type");
    // a key the dict lacks reads as None, as ptools' Val operator[] gives the reference (vmaf.cpp:83-86)
    static const PValPtr none = std::make_shared<PVal>();
    auto opt = [&](const char *k) -> const PValPtr & { const PValPtr *q = (*md)->find(k); return q ? *q : none; };
    const PValPtr &slopes = opt("slopes"), &intercepts = opt("intercepts");
    const PValPtr &score_clip = opt("score_clip"), &score_transform = opt("score_transform");

    // vmaf.cpp:95-136
    if (model_type->kind != PVal::STR || model_type->str != "LIBSVMNUSVR") {
        std::printf("Current vmafossexec only accepts model type LIBSVMNUSVR, but got %s\n", model_type->str.c_str());
        throw std::runtime_error("Incompatible model_type");
    }
    if (feature_names->kind != PVal::LIST) throw std::runtime_error("Incompatible feature_names");
    const bool is_none = norm_type->kind == PVal::STR && norm_type->str == "none";
    const bool is_lin = norm_type->kind == PVal::STR && norm_type->str == "linear_rescale";
    if (!is_none && !is_lin) throw std::runtime_error("Incompatible norm_type");
    if (is_lin && (slopes->kind != PVal::LIST || intercepts->kind != PVal::LIST)) throw std::runtime_error("Incompatible slopes or intercepts");
    if (!(score_clip->kind == PVal::NONE || score_clip->kind == PVal::LIST)) throw std::runtime_error("Incompatible score_clip");
    if (!(score_transform->kind == PVal::NONE || score_transform->kind == PVal::DICT)) throw std::runtime_error("Incompatible score_transform");

    auto num_of = [](const PValPtr &v, const char *what) {
        if (v->kind != PVal::NUM) throw std::runtime_error(std::string("Incompatible ") + what);
        return v->num;
    };
    for (auto &v : feature_names->list) {
        if (v->kind != PVal::STR) throw std::runtime_error("Incompatible feature_names");
        m.feature_names.push_back(v->str);
    }
    if (m.feature_names.empty()) throw std::runtime_error("Incompatible feature_names");
    m.linear_rescale = is_lin;
    if (is_lin) {
        for (auto &v : slopes->list) m.slopes.push_back(num_of(v, "slopes or intercepts"));
        for (auto &v : intercepts->list) m.intercepts.push_back(num_of(v, "slopes or intercepts"));
        if (m.slopes.size() < m.feature_names.size() + 1 || m.intercepts.size() < m.feature_names.size() + 1)
            throw std::runtime_error("Incompatible slopes or intercepts");
    }
    if (score_clip->kind == PVal::LIST) {
        if (score_clip->list.size() < 2) throw std::runtime_error("Incompatible score_clip");
        m.has_clip = true;
        m.clip_lo = num_of(score_clip->list[0], "score_clip");
        m.clip_hi = num_of(score_clip->list[1], "score_clip");
    }
    if (score_transform->kind == PVal::DICT) {
        ScoreTransform &t = m.transform;
        t.present = true;
        auto num = [&](const char *k, bool &has, double &v) {
            const PValPtr *p = score_transform->find(k);
            if (p && (*p)->kind == PVal::NUM) { has = true; v = (*p)->num; }
        };
        num("p0", t.has_p0, t.p0);
        num("p1", t.has_p1, t.p1);
        num("p2", t.has_p2, t.p2);
        auto flag = [&](const char *k) {
            const PValPtr *p = score_transform->find(k);
            return p && (*p)->kind == PVal::STR && (*p)->str == "true";
        };
        t.out_lte_in = flag("out_lte_in");
        t.out_gte_in = flag("out_gte_in");
    }
    load_svm_text(pkl_path + ".model", m);                 // vmaf.h:160-166: "<pkl>.model"
    if (m.dim > (int)m.feature_names.size()) throw std::runtime_error("Error loading SVM model");
    m.dim = (int)m.feature_names.size();
    {   // re-pack SV rows to the model's feature count (trailing omitted indices are zero)
        std::vector<double> sv((size_t)m.n_sv * m.dim, 0.0);
        const int old_dim = (int)(m.sv.size() / m.n_sv);
        for (int i = 0; i < m.n_sv; ++i)
            for (int j = 0; j < old_dim; ++j) sv[(size_t)i * m.dim + j] = m.sv[(size_t)i * old_dim + j];
        m.sv.swap(sv);
    }
    return m;
}

}  // namespace vmaf_amd
