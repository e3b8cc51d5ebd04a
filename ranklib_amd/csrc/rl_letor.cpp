// rl_letor.cpp -- host-side LETOR text parser behind the C ABI (include/rlhip.h: rl_letor_*).
//
// RankLib reads `label qid:ID fid:val fid:val ... # description` lines with String.split / Float.parseFloat
// (learning/DataPoint.java:58-110, features/FeatureManager.java:199-235).  The host mirror in ranklib_amd/learning.py follows that
// line for line in Python -- 6 k lines/s, ten minutes for an MSLR-WEB30K fold.  This parser does the common case natively on all
// host threads (two passes over the text: line index + largest feature id, then the values straight into the dense row matrix)
// and FLAGS every line that is not plain `number qid:token (digits:number)*` so that the Python code parses -- and, where the
// reference would, rejects -- exactly those lines itself.  Values are parsed by strtof: decimal -> float correctly rounded,
// as Float.parseFloat does.
#include <algorithm>
#include <cmath>
#include <cstdint>
#include <cstdlib>
#include <cstring>
#include <limits>
#include <string>
#include <thread>
#include <vector>

#include "../../include/rlhip.h"
#include "rl_internal.h"

struct rl_letor {
    const char *text = nullptr;
    int64_t len = 0;
    std::vector<int64_t> line_off; std::vector<int32_t> line_len;       // trimmed content of every data line (no '#' lines, no empty ones)
    std::vector<float> labels;
    std::vector<int32_t> last_fid;
    std::vector<int64_t> qid_off; std::vector<int32_t> qid_len;
    std::vector<int64_t> desc_off; std::vector<int32_t> desc_len;       // from '#' to the end of the trimmed line (len 0: none)
    std::vector<uint8_t> slow;                                          // 1: not the plain grammar, the caller parses this line itself
    int32_t max_fid = 0;
};

namespace {

inline bool is_space(char c) { return c == ' ' || c == '\t' || c == '\r' || c == '\n' || c == '\f' || c == '\v'; }

// [+-]?digits[.digits]?([eE][+-]?digits)?  or  [+-]?.digits...   -- the subset every implementation agrees on
inline bool plain_number(const char *p, const char *e)
{
    if (p < e && (*p == '+' || *p == '-')) p++;
    int nd = 0;
    while (p < e && *p >= '0' && *p <= '9') { p++; nd++; }
    if (p < e && *p == '.') { p++; while (p < e && *p >= '0' && *p <= '9') { p++; nd++; } }
    if (nd == 0) return false;
    if (p < e && (*p == 'e' || *p == 'E')) {
        p++;
        if (p < e && (*p == '+' || *p == '-')) p++;
        int ne = 0;
        while (p < e && *p >= '0' && *p <= '9') { p++; ne++; }
        if (ne == 0) return false;
    }
    return p == e;
}

inline float parse_float(const char *p, const char *e)
{
    char buf[64];
    const size_t n = (size_t)(e - p);
    if (n < sizeof(buf)) { memcpy(buf, p, n); buf[n] = 0; return strtof(buf, nullptr); }
    std::string s(p, e);
    return strtof(s.c_str(), nullptr);
}

struct LineView { const char *b, *e; };      // content before '#', trimmed

// pass 1 of a line: label, qid, description, largest feature id; false = needs the slow path
bool scan_line(const char *b, const char *e, float &label, const char *&qb, const char *&qe, int32_t &last, const char *&tok2)
{
    const char *p = b;
    const char *t0 = p; while (p < e && !is_space(*p)) p++;
    if (!plain_number(t0, p)) return false;
    label = parse_float(t0, p);
    if (!(label >= 0.0f)) return false;                               // negative labels are an error in the reference: let the caller raise it
    while (p < e && is_space(*p)) p++;
    if (p >= e) return false;
    const char *t1 = p; while (p < e && !is_space(*p)) p++;
    const char *colon = nullptr;
    for (const char *c = t1; c < p; c++) if (*c == ':') colon = c;     // text after the LAST ':' (getValue)
    qb = colon ? colon + 1 : t1; qe = p;
    while (p < e && is_space(*p)) p++;
    tok2 = p;
    last = 0;
    while (p < e) {
        const char *t = p; while (p < e && !is_space(*p)) p++;
        const char *c1 = t; while (c1 < p && *c1 != ':') c1++;
        if (c1 == p || c1 == t || c1 - t > 9) return false;
        int32_t fid = 0;
        for (const char *d = t; d < c1; d++) { if (*d < '0' || *d > '9') return false; fid = fid * 10 + (*d - '0'); }
        if (fid <= 0) return false;
        const char *c2 = p - 1; while (*c2 != ':') c2--;                // value = text after the LAST ':'
        if (!plain_number(c2 + 1, p)) return false;
        last = std::max(last, fid);
        while (p < e && is_space(*p)) p++;
    }
    return true;
}

void fill_row(const char *p, const char *e, float *row)
{
    while (p < e) {
        const char *t = p; while (p < e && !is_space(*p)) p++;
        const char *c1 = t; while (*c1 != ':') c1++;
        int32_t fid = 0;
        for (const char *d = t; d < c1; d++) fid = fid * 10 + (*d - '0');
        const char *c2 = p - 1; while (*c2 != ':') c2--;
        row[fid] = parse_float(c2 + 1, p);
        while (p < e && is_space(*p)) p++;
    }
}

template <class F> void parallel_for(int64_t n, F fn)
{
    const unsigned hw = std::max(1u, std::thread::hardware_concurrency());
    const int64_t nt = std::max<int64_t>(1, std::min<int64_t>(std::min<int64_t>(hw, 64), n / 2048 + 1));
    if (nt == 1) { fn(0, n); return; }
    std::vector<std::thread> th;
    for (int64_t k = 0; k < nt; k++) th.emplace_back([=]() { fn(n * k / nt, n * (k + 1) / nt); });
    for (auto &t : th) t.join();
}

}  // namespace

extern "C" {

int rl_letor_parse(const char *text, int64_t len, rl_letor **out)
{
    if (!text || !out || len < 0) return rl::fail(RL_ERR_INVALID, "null argument");
    *out = nullptr;
    rl_letor *L = new rl_letor();
    L->text = text; L->len = len;
    // line index: FeatureManager.readInput trims every line and skips empty ones and '#' comments (:203-206)
    for (int64_t p = 0; p < len;) {
        int64_t q = p;
        while (q < len && text[q] != '\n') q++;
        int64_t b = p, e = q;
        while (b < e && is_space(text[b])) b++;
        while (e > b && is_space(text[e - 1])) e--;
        if (e > b && text[b] != '#') { L->line_off.push_back(b); L->line_len.push_back((int32_t)std::min<int64_t>(e - b, 2147483647)); }
        p = q + 1;
    }
    const int64_t n = (int64_t)L->line_off.size();
    L->labels.assign(n, 0.f); L->last_fid.assign(n, 0); L->qid_off.assign(n, 0); L->qid_len.assign(n, 0);
    L->desc_off.assign(n, 0); L->desc_len.assign(n, 0); L->slow.assign(n, 0);
    parallel_for(n, [&](int64_t i0, int64_t i1) {
        for (int64_t i = i0; i < i1; i++) {
            const char *b = text + L->line_off[i], *e = b + L->line_len[i];
            const char *h = (const char *)memchr(b, '#', (size_t)(e - b));        // DataPoint.java:62-66
            const char *ce = e;
            if (h) { L->desc_off[i] = h - text; L->desc_len[i] = (int32_t)(e - h); ce = h; while (ce > b && is_space(ce[-1])) ce--; }
            float lab = 0.f; const char *qb = b, *qe = b, *t2 = b; int32_t last = 0;
            if (!scan_line(b, ce, lab, qb, qe, last, t2)) { L->slow[i] = 1; continue; }
            L->labels[i] = lab; L->last_fid[i] = last; L->qid_off[i] = qb - text; L->qid_len[i] = (int32_t)(qe - qb);
        }
    });
    int32_t mf = 0;
    for (int64_t i = 0; i < n; i++) mf = std::max(mf, L->last_fid[i]);
    L->max_fid = mf;
    *out = L;
    return RL_OK;
}

int rl_letor_info(const rl_letor *L, int64_t *n_docs, int32_t *max_fid, int64_t *n_slow)
{
    if (!L) return rl::fail(RL_ERR_INVALID, "null argument");
    if (n_docs) *n_docs = (int64_t)L->line_off.size();
    if (max_fid) *max_fid = L->max_fid;
    if (n_slow) { int64_t s = 0; for (uint8_t v : L->slow) s += v; *n_slow = s; }
    return RL_OK;
}

int rl_letor_arrays(const rl_letor *L, float *labels, int32_t *last_fid, int64_t *qid_off, int32_t *qid_len, int64_t *desc_off, int32_t *desc_len,
                    int64_t *line_off, int32_t *line_len, uint8_t *slow)
{
    if (!L) return rl::fail(RL_ERR_INVALID, "null argument");
    const size_t n = L->line_off.size();
    if (labels) memcpy(labels, L->labels.data(), n * sizeof(float));
    if (last_fid) memcpy(last_fid, L->last_fid.data(), n * sizeof(int32_t));
    if (qid_off) memcpy(qid_off, L->qid_off.data(), n * sizeof(int64_t));
    if (qid_len) memcpy(qid_len, L->qid_len.data(), n * sizeof(int32_t));
    if (desc_off) memcpy(desc_off, L->desc_off.data(), n * sizeof(int64_t));
    if (desc_len) memcpy(desc_len, L->desc_len.data(), n * sizeof(int32_t));
    if (line_off) memcpy(line_off, L->line_off.data(), n * sizeof(int64_t));
    if (line_len) memcpy(line_len, L->line_len.data(), n * sizeof(int32_t));
    if (slow) memcpy(slow, L->slow.data(), n);
    return RL_OK;
}

int rl_letor_rows(const rl_letor *L, float *X, int64_t row_stride)
{
    if (!L || !X || row_stride < (int64_t)L->max_fid + 1) return rl::fail(RL_ERR_INVALID, "rows: bad argument");
    const int64_t n = (int64_t)L->line_off.size();
    const float qnan = std::numeric_limits<float>::quiet_NaN();            // DenseDataPoint: unspecified values are UNKNOWN (NaN)
    parallel_for(n, [&](int64_t i0, int64_t i1) {
        for (int64_t i = i0; i < i1; i++) {
            float *row = X + i * row_stride;
            for (int64_t f = 0; f < row_stride; f++) row[f] = qnan;
            if (L->slow[i]) continue;
            const char *b = L->text + L->line_off[i], *e = b + L->line_len[i];
            if (L->desc_len[i] > 0) { e = L->text + L->desc_off[i]; while (e > b && is_space(e[-1])) e--; }
            const char *p = L->text + L->qid_off[i] + L->qid_len[i];         // the feature tokens follow the qid token
            while (p < e && is_space(*p)) p++;
            fill_row(p, e, row);
        }
    });
    return RL_OK;
}

void rl_letor_destroy(rl_letor *L) { delete L; }

}  // extern "C"
