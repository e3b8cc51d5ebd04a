// rl_model.cpp -- RankLib <ensemble> model text, written and parsed without a JVM.
#include "rl_model.h"

#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>

namespace rl {

// Shortest digit string d1d2..dn and decimal exponent e (value = 0.d1d2..dn x 10^e) that parses back to v.
template <typename T>
static void shortest_digits(T v, std::string &digits, int &exp10)
{
    char buf[64];
    const int maxp = sizeof(T) == 4 ? 9 : 17;
    for (int p = 1; p <= maxp; p++) {
        snprintf(buf, sizeof buf, "%.*e", p - 1, (double)v);
        const T back = sizeof(T) == 4 ? (T)strtof(buf, nullptr) : (T)strtod(buf, nullptr);
        if (back == v || p == maxp) break;
    }
    // buf = d.ddddde[+-]xx
    digits.clear();
    const char *e = strchr(buf, 'e');
    for (const char *c = buf; c < e; c++) if (*c >= '0' && *c <= '9') digits.push_back(*c);
    exp10 = atoi(e + 1) + 1;
    while (digits.size() > 1 && digits.back() == '0') digits.pop_back();
}

template <typename T>
static std::string java_to_string(T v)
{
    if (v != v) return "NaN";
    if (std::isinf((double)v)) return v > 0 ? "Infinity" : "-Infinity";
    if (v == 0) return std::signbit((double)v) ? "-0.0" : "0.0";
    std::string s;
    if (v < 0) { s = "-"; v = -v; }
    std::string d; int e;
    shortest_digits<T>(v, d, e);
    const double av = (double)v;
    if (av >= 1e-3 && av < 1e7) {
        if (e <= 0) { s += "0."; s.append((size_t)(-e), '0'); s += d; }
        else if ((int)d.size() <= e) { s += d; s.append((size_t)(e - (int)d.size()), '0'); s += ".0"; }
        else { s += d.substr(0, (size_t)e); s += "."; s += d.substr((size_t)e); }
    } else {
        s += d.substr(0, 1); s += ".";
        s += d.size() > 1 ? d.substr(1) : std::string("0");
        s += "E"; s += std::to_string(e - 1);
    }
    return s;
}

std::string java_float_to_string(float v) { return java_to_string<float>(v); }
std::string java_double_to_string(double v) { return java_to_string<double>(v); }

// Split.getString  learning/tree/Split.java:140-155
static void write_node(const HostTree &t, int n, const std::string &indent, std::string &out)
{
    if (t.feature[n] == -1) {
        out += indent + "<output>" + java_double_to_string((double)t.output[n]) + " </output>\n";
        return;
    }
    out += indent + "<feature>" + std::to_string(t.feature[n]) + " </feature>\n";
    out += indent + "<threshold> " + java_float_to_string(t.threshold[n]) + " </threshold>\n";
    out += indent + "<split pos=\"left\">\n";
    write_node(t, t.left[n], indent + "\t", out);
    out += indent + "</split>\n";
    out += indent + "<split pos=\"right\">\n";
    write_node(t, t.right[n], indent + "\t", out);
    out += indent + "</split>\n";
}

std::string model_to_text(const ModelHeader &h, const std::vector<HostTree> &trees)
{
    std::string o;
    o += std::string("## ") + h.name + "\n";                                                        // LambdaMART.java:292
    o += "## No. of trees = " + std::to_string(h.n_trees) + "\n";
    o += "## No. of leaves = " + std::to_string(h.n_leaves) + "\n";
    o += "## No. of threshold candidates = " + std::to_string(h.n_threshold) + "\n";
    o += "## Learning rate = " + java_float_to_string(h.learning_rate) + "\n";
    o += "## Stop early = " + std::to_string(h.early_stop) + "\n";
    o += "\n";
    o += "<ensemble>\n";                                                             // Ensemble.toString :119-130
    for (size_t i = 0; i < trees.size(); i++) {
        o += "\t<tree id=\"" + std::to_string(i + 1) + "\" weight=\"" + java_float_to_string(trees[i].weight) + "\">\n";
        o += "\t\t<split>\n";                                                        // Split.toString(indent) :132-138
        write_node(trees[i], 0, "\t\t\t", o);
        o += "\t\t</split>\n";
        o += "\t</tree>\n";
    }
    o += "</ensemble>\n";
    return o;
}

// ---- parser ------------------------------------------------------------------------------------------
namespace {
struct Parser {
    const std::string &s; size_t p = 0; std::string err;
    explicit Parser(const std::string &t) : s(t) {}
    void ws() { while (p < s.size() && (s[p] == ' ' || s[p] == '\t' || s[p] == '\n' || s[p] == '\r')) p++; }
    bool peek_tag(const char *name)
    {   // next non-space token is "<name" followed by space, '>' ?
        ws();
        const size_t n = strlen(name);
        return p + 1 + n <= s.size() && s[p] == '<' && s.compare(p + 1, n, name) == 0 &&
               (p + 1 + n == s.size() || s[p + 1 + n] == '>' || s[p + 1 + n] == ' ' || s[p + 1 + n] == '\t');
    }
    bool open(const char *name, std::string *attrs = nullptr)
    {
        if (!peek_tag(name)) { err = std::string("expected <") + name + "> at offset " + std::to_string(p); return false; }
        const size_t e = s.find('>', p);
        if (e == std::string::npos) { err = "unterminated tag"; return false; }
        if (attrs) *attrs = s.substr(p + 1 + strlen(name), e - p - 1 - strlen(name));
        p = e + 1;
        return true;
    }
    bool close(const char *name)
    {
        ws();
        const std::string t = std::string("</") + name + ">";
        if (s.compare(p, t.size(), t) != 0) { err = "expected " + t + " at offset " + std::to_string(p); return false; }
        p += t.size();
        return true;
    }
    bool text_until_close(const char *name, std::string &out)
    {
        const std::string t = std::string("</") + name + ">";
        const size_t e = s.find(t, p);
        if (e == std::string::npos) { err = "missing " + t; return false; }
        out = s.substr(p, e - p);
        p = e + t.size();
        // String.trim()
        size_t a = 0, b = out.size();
        while (a < b && (unsigned char)out[a] <= ' ') a++;
        while (b > a && (unsigned char)out[b - 1] <= ' ') b--;
        out = out.substr(a, b - a);
        return true;
    }
};

bool parse_float(const std::string &v, float &out)
{   // Float.parseFloat: decimal / scientific, optional f/F/d/D suffix, "NaN", "Infinity"
    std::string t = v;
    if (!t.empty() && (t.back() == 'f' || t.back() == 'F' || t.back() == 'd' || t.back() == 'D')) t.pop_back();
    if (t.empty()) return false;
    char *end = nullptr;
    out = strtof(t.c_str(), &end);
    return end && *end == 0;
}

// Ensemble.create  learning/tree/Ensemble.java:141-159 ; the cursor is just past "<split ...>"
bool parse_split_body(Parser &ps, HostTree &t, int &me)
{
    me = t.n_nodes++;
    t.feature.push_back(-1); t.threshold.push_back(0.f); t.left.push_back(-1); t.right.push_back(-1); t.output.push_back(0.f);
    t.deviance.push_back(0.0); t.count.push_back(0);
    if (ps.peek_tag("feature")) {
        std::string v;
        if (!ps.open("feature") || !ps.text_until_close("feature", v)) return false;
        char *end = nullptr;
        const long fid = strtol(v.c_str(), &end, 10);
        if (!end || *end != 0 || v.empty()) { ps.err = "bad feature id '" + v + "'"; return false; }
        t.feature[me] = (int32_t)fid;
        if (!ps.open("threshold") || !ps.text_until_close("threshold", v)) return false;
        float th;
        if (!parse_float(v, th)) { ps.err = "bad threshold '" + v + "'"; return false; }
        t.threshold[me] = th;
        int l, r;
        if (!ps.open("split") || !parse_split_body(ps, t, l) || !ps.close("split")) return false;
        if (!ps.open("split") || !parse_split_body(ps, t, r) || !ps.close("split")) return false;
        t.left[me] = l; t.right[me] = r;
        return true;
    }
    std::string v;
    if (!ps.open("output") || !ps.text_until_close("output", v)) return false;
    float o;
    if (!parse_float(v, o)) { ps.err = "bad output '" + v + "'"; return false; }
    t.output[me] = o;
    return true;
}
}  // namespace

bool model_from_text(const std::string &text, std::vector<HostTree> &trees, std::string &err)
{
    // ModelLineProducer: drop lines starting with "##", concatenate the rest without the line breaks
    std::string body;
    size_t i = 0;
    while (i < text.size()) {
        size_t e = text.find('\n', i);
        if (e == std::string::npos) e = text.size();
        size_t a = i;
        while (a < e && (text[a] == ' ' || text[a] == '\t' || text[a] == '\r')) a++;
        if (!(a < e && text[a] == '#')) { body.append(text, i, e - i); body.push_back('\n'); }
        i = e + 1;
    }
    Parser ps(body);
    trees.clear();
    if (!ps.open("ensemble")) { err = ps.err; return false; }
    while (ps.peek_tag("tree")) {
        std::string attrs;
        if (!ps.open("tree", &attrs)) { err = ps.err; return false; }
        HostTree t;
        const size_t w = attrs.find("weight=\"");
        if (w == std::string::npos) { err = "tree without weight"; return false; }
        const size_t we = attrs.find('"', w + 8);
        if (we == std::string::npos || !parse_float(attrs.substr(w + 8, we - w - 8), t.weight)) { err = "bad tree weight"; return false; }
        int root;
        if (!ps.open("split") || !parse_split_body(ps, t, root) || !ps.close("split") || !ps.close("tree")) { err = ps.err; return false; }
        trees.push_back(t);
    }
    if (!ps.close("ensemble")) { err = ps.err; return false; }
    return true;
}

}  // namespace rl
