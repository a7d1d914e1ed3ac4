// Wavefront OBJ reader behind nr.load_obj (host code only; no kernel).  Replaces the reference's four Python passes
// over `lines` (neural_renderer/load_obj.py:117-176) by one native pass over the file bytes.
//
// Semantics kept from the reference parser:
//   * a line's first whitespace-separated token selects it: `v` (first 3 numbers), `vn` (3), `vt` (first 2), `f`;
//     everything else is ignored here (mtllib / usemtl are handled by the Python side only when textures are loaded);
//   * `f` entries are `v`, `v/vt`, `v//vn` or `v/vt/vn`; the vertex index is field 0, the texcoord index field 1 —
//     read iff the FILE has `vt` lines (has_vt, load_obj.py:146,166), the normal index the LAST field — read iff the
//     file has `vn` lines (load_obj.py:133,169); indices are stored 0-based (`- 1`, load_obj.py:171-173), negative
//     (relative) indices are not resolved (the reference does not either);
//   * numbers are parsed as correctly rounded doubles (what Python's float() does) and narrowed to float32
//     (`.astype(np.float32)`).
// Faces must be triangles (the reference's np.vstack + downstream code assume [nf,3]); anything else is an error.
#include "rnr_internal.h"

#include <cmath>

#include <charconv>
#include <cstring>

namespace rnr {
namespace {

struct Cursor {
    const char* p;
    const char* end;
};

inline bool is_space(char c) { return c == ' ' || c == '\t' || c == '\r' || c == '\v' || c == '\f'; }

inline void skip_space(Cursor& c) {
    while (c.p < c.end && is_space(*c.p)) c.p++;
}

// next whitespace-separated token of the current line; false at end of line
inline bool next_token(Cursor& c, const char*& tb, const char*& te) {
    skip_space(c);
    if (c.p >= c.end || *c.p == '\n') return false;
    tb = c.p;
    while (c.p < c.end && *c.p != '\n' && !is_space(*c.p)) c.p++;
    te = c.p;
    return true;
}

inline void skip_line(Cursor& c) {
    while (c.p < c.end && *c.p != '\n') c.p++;
    if (c.p < c.end) c.p++;
}

inline bool parse_float(const char* b, const char* e, float& out) {
    if (b < e && *b == '+') {                    // float('+1.5') is legal Python; from_chars rejects the sign
        b++;
        if (b < e && (*b == '+' || *b == '-')) return false;           // '+-1': ValueError in Python
    }
    if (e > b && e[-1] == ')') return false;     // from_chars reads 'nan(chars)'; float() does not
    double d = 0.0;
    auto r = std::from_chars(b, e, d, std::chars_format::general);
    if (r.ec == std::errc::result_out_of_range && r.ptr == e) {
        // a well-formed decimal beyond double's range: float('1e400') = inf, float('1e-400') = 0.0 — strtod's answers
        char buf[64];
        const size_t n = (size_t)(e - b);
        if (n >= sizeof(buf)) return false;
        memcpy(buf, b, n);
        buf[n] = 0;
        d = strtod(buf, nullptr);
    } else if (r.ec != std::errc() || r.ptr != e) {
        // What else Python's float() takes: [sign] inf | infinity | nan, any case.  Nothing more — strtod would also accept
        // hexadecimal ('0x10' -> 16.0) and 'nan(...)' where the reference parser (load_obj.py:120-135, float()) raises
        // ValueError.  Known deviations: Python's underscore literals ('1_0') and non-ASCII digits are rejected here.
        const char* q = b;
        bool neg = false;
        if (q < e && *q == '-') { neg = true; q++; }      // (a leading '+' was consumed above)
        char w[9];
        const size_t n = (size_t)(e - q);
        if (n == 0 || n > 8) return false;
        for (size_t i = 0; i < n; i++) w[i] = (char)(q[i] | 0x20);
        w[n] = 0;
        if (!strcmp(w, "inf") || !strcmp(w, "infinity")) d = neg ? -HUGE_VAL : HUGE_VAL;
        else if (!strcmp(w, "nan")) d = nan("");
        else return false;
    }
    out = (float)d;
    return true;
}

inline bool parse_int(const char* b, const char* e, int32_t& out) {
    if (b < e && *b == '+') b++;
    long v = 0;
    auto r = std::from_chars(b, e, v);
    if (r.ec != std::errc() || r.ptr != e) return false;
    out = (int32_t)v;
    return true;
}

enum { LINE_OTHER = 0, LINE_V, LINE_VN, LINE_VT, LINE_F };

inline int classify(const char* tb, const char* te) {
    const size_t n = (size_t)(te - tb);
    if (n == 1 && tb[0] == 'v') return LINE_V;
    if (n == 1 && tb[0] == 'f') return LINE_F;
    if (n == 2 && tb[0] == 'v' && tb[1] == 'n') return LINE_VN;
    if (n == 2 && tb[0] == 'v' && tb[1] == 't') return LINE_VT;
    return LINE_OTHER;
}

long line_number(const char* text, const char* at) {
    long n = 1;
    for (const char* q = text; q < at; q++) n += *q == '\n';
    return n;
}

}  // namespace
}  // namespace rnr

using namespace rnr;

extern "C" int rnr_obj_scan(const char* text, size_t len, rnr_obj_counts* counts) {
    RNR_REQUIRE(text && counts, "rnr_obj_scan: null argument");
    memset(counts, 0, sizeof(*counts));
    Cursor c{text, text + len};
    while (c.p < c.end) {
        const char *tb, *te;
        if (next_token(c, tb, te)) {
            switch (classify(tb, te)) {
                case LINE_V: counts->num_vertices++; break;
                case LINE_VN: counts->num_normals++; break;
                case LINE_VT: counts->num_texcoords++; break;
                case LINE_F: counts->num_faces++; break;
                default: break;
            }
        }
        skip_line(c);
    }
    return 0;
}

extern "C" int rnr_obj_parse(const char* text, size_t len, const rnr_obj_counts* counts, float* v, float* vn, float* vt,
                             int32_t* f_v_idx, int32_t* f_vt_idx, int32_t* f_vn_idx) {
    RNR_REQUIRE(text && counts, "rnr_obj_parse: null argument");
    RNR_REQUIRE((v || !counts->num_vertices) && (vn || !counts->num_normals) && (vt || !counts->num_texcoords) &&
                (f_v_idx || !counts->num_faces), "rnr_obj_parse: an output array is missing");
    const bool has_vt = counts->num_texcoords > 0, has_vn = counts->num_normals > 0;
    RNR_REQUIRE(!counts->num_faces || ((f_vt_idx || !has_vt) && (f_vn_idx || !has_vn)), "rnr_obj_parse: an index array is missing");
    long iv = 0, ivn = 0, ivt = 0, ifc = 0;
    Cursor c{text, text + len};
    while (c.p < c.end) {
        const char* line = c.p;
        const char *tb, *te;
        if (!next_token(c, tb, te)) { skip_line(c); continue; }
        const int kind = classify(tb, te);
        if (kind == LINE_V || kind == LINE_VN || kind == LINE_VT) {
            const int want = kind == LINE_VT ? 2 : 3;
            float* dst = kind == LINE_V ? v + 3 * iv : (kind == LINE_VN ? vn + 3 * ivn : vt + 2 * ivt);
            const long have = kind == LINE_V ? iv : (kind == LINE_VN ? ivn : ivt);
            const long cap = kind == LINE_V ? counts->num_vertices : (kind == LINE_VN ? counts->num_normals : counts->num_texcoords);
            RNR_REQUIRE(have < cap, "rnr_obj_parse: more '%.*s' lines than rnr_obj_scan counted", (int)(te - tb), tb);
            for (int k = 0; k < want; k++) {
                const char *nb, *ne;
                RNR_REQUIRE(next_token(c, nb, ne), "OBJ line %ld: '%.*s' needs %d numbers", line_number(text, line),
                            (int)(te - tb), tb, want);
                RNR_REQUIRE(parse_float(nb, ne, dst[k]), "OBJ line %ld: cannot parse number '%.*s'", line_number(text, line),
                            (int)(ne - nb), nb);
            }
            if (kind == LINE_V) iv++; else if (kind == LINE_VN) ivn++; else ivt++;
        } else if (kind == LINE_F) {
            RNR_REQUIRE(ifc < counts->num_faces, "rnr_obj_parse: more 'f' lines than rnr_obj_scan counted");
            int nvert = 0;
            const char *eb, *ee;
            while (next_token(c, eb, ee)) {
                RNR_REQUIRE(nvert < 3, "OBJ line %ld: only triangles are supported (face with more than 3 vertices)",
                            line_number(text, line));
                // split the entry at '/': field 0, field 1, last field
                const char* s1 = (const char*)memchr(eb, '/', (size_t)(ee - eb));
                const char* f0e = s1 ? s1 : ee;
                int32_t idx;
                RNR_REQUIRE(parse_int(eb, f0e, idx), "OBJ line %ld: bad vertex index '%.*s'", line_number(text, line),
                            (int)(ee - eb), eb);
                f_v_idx[3 * ifc + nvert] = idx - 1;
                if (has_vt) {
                    RNR_REQUIRE(s1, "OBJ line %ld: face entry '%.*s' has no texcoord index although the file has vt lines",
                                line_number(text, line), (int)(ee - eb), eb);
                    const char* s2 = (const char*)memchr(s1 + 1, '/', (size_t)(ee - s1 - 1));
                    RNR_REQUIRE(parse_int(s1 + 1, s2 ? s2 : ee, idx), "OBJ line %ld: bad texcoord index in '%.*s'",
                                line_number(text, line), (int)(ee - eb), eb);
                    f_vt_idx[3 * ifc + nvert] = idx - 1;
                }
                if (has_vn) {
                    const char* last = eb;
                    for (const char* q = eb; q < ee; q++)
                        if (*q == '/') last = q + 1;
                    RNR_REQUIRE(parse_int(last, ee, idx), "OBJ line %ld: bad normal index in '%.*s'", line_number(text, line),
                                (int)(ee - eb), eb);
                    f_vn_idx[3 * ifc + nvert] = idx - 1;
                }
                nvert++;
            }
            RNR_REQUIRE(nvert == 3, "OBJ line %ld: only triangles are supported (face with %d vertices)",
                        line_number(text, line), nvert);
            ifc++;
        }
        skip_line(c);
    }
    RNR_REQUIRE(iv == counts->num_vertices && ivn == counts->num_normals && ivt == counts->num_texcoords &&
                ifc == counts->num_faces, "rnr_obj_parse: counts do not match the text (scan and parse saw different input)");
    return 0;
}
