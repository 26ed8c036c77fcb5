// scene_parser.cpp -- librt_host.so: the scene-file front end (include/rt_host.h, rth_scene_*).
//
// Mirrors the token loops of the reference's `main` (HW4/hw4/hw4.cpp:538-765, Final/photon/final.cpp:509-735):
// whitespace-separated tokens, a keyword (E V F R S T L ML M O RectL) followed by a fixed number of numeric
// tokens read with atof / atoi semantics, unknown tokens skipped.  What it produces is the flat arrays of
// rt_scene_desc, i.e. what the reference leaves in simpleMeshes / simpleLights / simpleObjects / camera.
//
// Built for large inputs (10 M triangles are ~1.3 GB of text): the file is memory-mapped, cut at keyword
// tokens into one chunk per thread, the chunks are parsed in parallel and stitched in file order (the material
// in force and the open object carry over a chunk boundary); numbers take an exact fast path (<= 15 significant
// digits, |exponent| <= 22: one correctly rounded double operation, the value strtod returns) and fall back to
// strtod otherwise.  A binary cache (rth_scene_save / rth_scene_load) skips the text altogether.
#include "../../include/rt_host.h"

#include <fcntl.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>

#include <algorithm>
#include <array>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <map>
#include <string>
#include <thread>
#include <vector>

struct rth_scene {
    int flavour = RTH_FLAVOUR_HW4;
    std::vector<float> prims;            // 12 per primitive (rt_prim)
    std::vector<uint8_t> type;
    std::vector<int32_t> mat;
    std::vector<float> materials;        // 10 per material (rt_material)
    std::vector<int64_t> obj_first, obj_count;
    std::vector<rt_light> lights;
    float eye[3] = {0, 0, 0}, view[3] = {0, 0, 1}, fov = 0.f;
    int32_t resx = 0, resy = 0;
};

namespace {

inline bool is_space(char c) { return c == ' ' || c == '\n' || c == '\t' || c == '\r' || c == '\v' || c == '\f'; }
inline bool is_alpha(char c) { return (c >= 'A' && c <= 'Z') || (c >= 'a' && c <= 'z'); }

struct Tok { const char* b; const char* e; };

struct Cursor {
    const char* p; const char* end;
    bool next(Tok& t) {
        while (p < end && is_space(*p)) ++p;
        if (p >= end) { t.b = t.e = end; return false; }
        t.b = p;
        while (p < end && !is_space(*p)) ++p;
        t.e = p;
        return true;
    }
};

const double kPow10[23] = {1e0, 1e1, 1e2, 1e3, 1e4, 1e5, 1e6, 1e7, 1e8, 1e9, 1e10, 1e11, 1e12, 1e13, 1e14, 1e15, 1e16,
                           1e17, 1e18, 1e19, 1e20, 1e21, 1e22};

double slow_atof(const Tok& t) {
    char buf[128];
    size_t n = (size_t)(t.e - t.b);
    if (n >= sizeof buf) n = sizeof buf - 1;
    memcpy(buf, t.b, n); buf[n] = 0;
    return atof(buf);
}

// atof(): the longest numeric prefix, 0.0 when there is none
double tok_atof(const Tok& t) {
    const char* p = t.b; const char* e = t.e;
    if (p >= e) return 0.0;
    bool neg = false;
    if (*p == '+' || *p == '-') { neg = (*p == '-'); ++p; }
    if (p >= e || !((*p >= '0' && *p <= '9') || *p == '.')) return slow_atof(t);     // nan, inf, junk: libc decides
    uint64_t mant = 0; int digits = 0, exp10 = 0; bool any = false;
    while (p < e && *p >= '0' && *p <= '9') {
        any = true;
        if (digits < 19) { mant = mant * 10 + (uint64_t)(*p - '0'); if (mant) ++digits; } else ++exp10;
        ++p;
    }
    if (p < e && *p == '.') {
        ++p;
        while (p < e && *p >= '0' && *p <= '9') {
            any = true;
            if (digits < 19) { mant = mant * 10 + (uint64_t)(*p - '0'); if (mant) ++digits; --exp10; }
            ++p;
        }
    }
    if (!any) return 0.0;
    if (p < e && (*p == 'e' || *p == 'E')) {
        const char* q = p + 1; bool eneg = false;
        if (q < e && (*q == '+' || *q == '-')) { eneg = (*q == '-'); ++q; }
        if (q < e && *q >= '0' && *q <= '9') {
            int ex = 0;
            while (q < e && *q >= '0' && *q <= '9') { if (ex < 100000) ex = ex * 10 + (*q - '0'); ++q; }
            exp10 += eneg ? -ex : ex;
            p = q;
        }
    }
    if (p < e && (*p == 'x' || *p == 'X')) return slow_atof(t);                      // hex float
    if (digits <= 15 && exp10 >= -22 && exp10 <= 22) {                                // exact operands, one rounding
        double d = (double)mant;
        d = exp10 < 0 ? d / kPow10[-exp10] : d * kPow10[exp10];
        return neg ? -d : d;
    }
    return slow_atof(t);
}

int tok_atoi(const Tok& t) {
    const char* p = t.b; bool neg = false; long long v = 0;
    if (p < t.e && (*p == '+' || *p == '-')) { neg = (*p == '-'); ++p; }
    while (p < t.e && *p >= '0' && *p <= '9') { if (v < (1ll << 40)) v = v * 10 + (*p - '0'); ++p; }
    v = neg ? -v : v;
    return (int)std::max<long long>(std::min<long long>(v, 2147483647ll), -2147483647ll - 1);
}

inline bool tok_is(const Tok& t, const char* s) {
    const size_t n = strlen(s);
    return (size_t)(t.e - t.b) == n && memcmp(t.b, s, n) == 0;
}

// vec3::normalize, algebra3.cpp:280-282,295-302: length in float, d_inv = float(1. / length)
inline void normalize_f32(float* v) {
    const float len = std::sqrt((v[0] * v[0] + v[1] * v[1]) + v[2] * v[2]);
    const float di = (float)(1.0 / (double)len);
    v[0] *= di; v[1] *= di; v[2] *= di;
}

struct Chunk {
    std::vector<float> prims;
    std::vector<uint8_t> type;
    struct Run { int64_t start; int mat_local; };        // mat_local -1: the material in force before this chunk
    std::vector<Run> runs;
    std::vector<std::array<float, 10>> mats;              // the chunk's M lines, in order
    std::vector<int64_t> o_marks;                         // chunk-local primitive count at every `O`
    std::vector<rt_light> lights;
    bool has_e = false, has_v = false, has_f = false, has_r = false;
    float eye[3], view[3], fov; int resx, resy;
};

rt_light point_light(const float* pos, const float* col, float intensity) {
    rt_light l; memset(&l, 0, sizeof l);
    l.type = RT_LIGHT_POINT;
    for (int k = 0; k < 3; ++k) { l.pos[k] = pos[k]; l.color[k] = col[k]; }
    l.intensity = intensity;
    l.dir1[0] = 1.f; l.dir2[2] = 1.f; l.n1 = l.n2 = 1;
    return l;
}

// Parses the statements that START in [b, e): a statement's arguments are read to the end of the text (`file_end`), whatever
// they look like -- the reference reads a fixed number of tokens after a keyword (hw4.cpp:538-747).  Returns where the next
// statement would start: the next chunk's cut, if the cut was a true statement start.
const char* parse_chunk(const char* b, const char* e, const char* file_end, int flavour, Chunk& c) {
    c = Chunk();
    Cursor cur{b, file_end};
    Tok t;
    int mat_now = -1;
    bool run_open = false;
    auto num = [&]() -> double { Tok a; cur.next(a); return tok_atof(a); };
    auto inum = [&]() -> int { Tok a; cur.next(a); return tok_atoi(a); };
    auto add_prim = [&](const float* v12, uint8_t ty) {
        if (!run_open) { c.runs.push_back({(int64_t)c.type.size(), mat_now}); run_open = true; }
        c.prims.insert(c.prims.end(), v12, v12 + 12);
        c.type.push_back(ty);
    };
    for (;;) {
        if (!cur.next(t)) return file_end;
        if (t.b >= e) return t.b;
        if (tok_is(t, "T")) {
            float v[12];
            for (int k = 0; k < 9; ++k) v[k] = (float)num();
            if (flavour == RTH_FLAVOUR_HW4) {
                for (int k = 9; k < 12; ++k) v[k] = (float)num();
            } else {                                           // final.cpp:608: (v2 - v0) ^ (v1 - v0)
                const float a[3] = {v[6] - v[0], v[7] - v[1], v[8] - v[2]}, d[3] = {v[3] - v[0], v[4] - v[1], v[5] - v[2]};
                v[9] = a[1] * d[2] - a[2] * d[1]; v[10] = a[2] * d[0] - a[0] * d[2]; v[11] = a[0] * d[1] - a[1] * d[0];
            }
            normalize_f32(v + 9);
            add_prim(v, RT_PRIM_TRIANGLE);
        } else if (tok_is(t, "S")) {
            float v[12] = {0};
            for (int k = 0; k < 4; ++k) v[k] = (float)num();
            add_prim(v, RT_PRIM_SPHERE);
        } else if (tok_is(t, "M")) {
            std::array<float, 10> m;
            for (int k = 0; k < 10; ++k) m[k] = (float)num();
            c.mats.push_back(m); mat_now = (int)c.mats.size() - 1; run_open = false;
        } else if (tok_is(t, "O")) {
            if (flavour == RTH_FLAVOUR_HW4) c.o_marks.push_back((int64_t)c.type.size());
        } else if (tok_is(t, "E")) {
            for (int k = 0; k < 3; ++k) c.eye[k] = (float)num();
            c.has_e = true;
        } else if (tok_is(t, "V")) {
            for (int k = 0; k < 3; ++k) c.view[k] = (float)num();
            c.has_v = true;
        } else if (tok_is(t, "F")) {
            c.fov = (float)num(); c.has_f = true;
        } else if (tok_is(t, "R")) {
            c.resx = inum(); c.resy = inum(); c.has_r = true;
        } else if (tok_is(t, "L")) {
            float p[3]; const float w[3] = {1.f, 1.f, 1.f};
            for (int k = 0; k < 3; ++k) p[k] = (float)num();
            c.lights.push_back(point_light(p, w, 1.0f));
        } else if (tok_is(t, "ML")) {
            float v[7];
            for (int k = 0; k < 7; ++k) v[k] = (float)num();
            c.lights.push_back(point_light(v, v + 3, v[6]));
        } else if (tok_is(t, "RectL") && flavour == RTH_FLAVOUR_FINAL) {
            float p[3]; const float w[3] = {1.f, 1.f, 1.f};
            for (int k = 0; k < 3; ++k) p[k] = (float)num();
            rt_light l = point_light(p, w, 1.0f);
            l.type = RT_LIGHT_RECT;
            l.size1 = (float)num(); l.size2 = (float)num(); l.n1 = inum(); l.n2 = inum();
            c.lights.push_back(l);
        }
        // anything else: skipped, as the reference's chain of strcmp does
    }
}

int stitch(std::vector<Chunk>& chunks, int flavour, rth_scene* s) {
    s->flavour = flavour;
    int64_t total = 0;
    for (const Chunk& c : chunks) total += (int64_t)c.type.size();
    s->prims.resize((size_t)total * 12); s->type.resize((size_t)total); s->mat.resize((size_t)total);
    // materials: de-duplicated in order of first use; default of hw4.cpp:521-526 while no M line is in force
    const std::array<float, 10> def = {1.f, 1.f, 1.f, 1.f, 1.f, 1.f, 6.0f, 0.9f, 0.f, 1.0f};
    std::map<std::array<float, 10>, int> index;
    std::array<float, 10> in_force = def;
    auto mat_id = [&](const std::array<float, 10>& m) {
        auto it = index.find(m);
        if (it != index.end()) return it->second;
        const int id = (int)index.size();
        index.emplace(m, id);
        s->materials.insert(s->materials.end(), m.begin(), m.end());
        return id;
    };
    int64_t base = 0, obj_start = 0;
    bool has_o = false;
    std::vector<int64_t> bases;
    for (Chunk& c : chunks) {
        bases.push_back(base);
        const int64_t n = (int64_t)c.type.size();
        for (size_t r = 0; r < c.runs.size(); ++r) {
            const int64_t a = c.runs[r].start, b = r + 1 < c.runs.size() ? c.runs[r + 1].start : n;
            const int id = mat_id(c.runs[r].mat_local < 0 ? in_force : c.mats[(size_t)c.runs[r].mat_local]);
            std::fill(s->mat.begin() + base + a, s->mat.begin() + base + b, id);
        }
        if (!c.mats.empty()) in_force = c.mats.back();
        for (int64_t m : c.o_marks) {                            // hw4.cpp:729-765: an empty object is dropped
            const int64_t at = base + m;
            if (at - obj_start == 0) continue;
            has_o = true;
            s->obj_first.push_back(obj_start); s->obj_count.push_back(at - obj_start); obj_start = at;
        }
        if (c.has_e) memcpy(s->eye, c.eye, sizeof s->eye);
        if (c.has_v) memcpy(s->view, c.view, sizeof s->view);
        if (c.has_f) s->fov = c.fov;
        if (c.has_r) { s->resx = c.resx; s->resy = c.resy; }
        s->lights.insert(s->lights.end(), c.lights.begin(), c.lights.end());
        base += n;
    }
    // the bulk copy, chunk by chunk in parallel
    std::vector<std::thread> th;
    for (size_t i = 0; i < chunks.size(); ++i)
        th.emplace_back([&, i] {
            const Chunk& c = chunks[i];
            if (c.type.empty()) return;
            memcpy(s->prims.data() + (size_t)bases[i] * 12, c.prims.data(), c.prims.size() * sizeof(float));
            memcpy(s->type.data() + (size_t)bases[i], c.type.data(), c.type.size());
        });
    for (std::thread& t : th) t.join();
    if (flavour == RTH_FLAVOUR_HW4) {
        if (!has_o) {                                            // hw4.cpp:754-765: no `O` at all = one object
            if (total) { s->obj_first.push_back(0); s->obj_count.push_back(total); }
        } else {
            // meshes after the last `O` are parsed but never put into a tree (hw4.cpp:768): the reference cannot hit them
            for (int64_t i = obj_start; i < total; ++i) s->type[(size_t)i] = RT_PRIM_DISABLED;
        }
    }
    if (s->materials.empty()) s->materials.insert(s->materials.end(), def.begin(), def.end());
    return RT_OK;
}

int parse_range(const char* b, const char* e, int flavour, int nthreads, rth_scene** out) {
    if (flavour != RTH_FLAVOUR_HW4 && flavour != RTH_FLAVOUR_FINAL) return RT_E_INVALID;
    const int64_t len = (int64_t)(e - b);
    int nt = nthreads > 0 ? nthreads : (int)std::min<unsigned>(std::max(1u, std::thread::hardware_concurrency()), 16u);
    if (len < (1 << 20)) nt = 1;
    // cut points: a GUESS at a statement start (a token whose first character is a letter, after white space).  A chunk
    // reads its statements' arguments past its end, and reports where the next statement starts; where that is not the
    // next chunk's cut (an argument such as `nan` or `inf` was taken for a keyword) the next chunk is parsed again from
    // there, so the result is the single-threaded parse whatever nthreads is.
    std::vector<const char*> cut{b};
    for (int i = 1; i < nt; ++i) {
        const char* p = b + len * i / nt;
        if (p <= cut.back()) continue;
        while (p < e && !(is_alpha(*p) && is_space(p[-1]))) ++p;
        if (p < e && p > cut.back()) cut.push_back(p);
    }
    cut.push_back(e);
    std::vector<Chunk> chunks(cut.size() - 1);
    std::vector<const char*> stop(cut.size() - 1, e);
    std::vector<std::thread> th;
    for (size_t i = 0; i + 1 < cut.size(); ++i)
        th.emplace_back([&, i] { stop[i] = parse_chunk(cut[i], cut[i + 1], e, flavour, chunks[i]); });
    for (std::thread& t : th) t.join();
    for (size_t i = 1; i + 1 < cut.size(); ++i) {
        const char* from = stop[i - 1];
        if (from == cut[i]) continue;
        if (from >= cut[i + 1]) { chunks[i] = Chunk(); stop[i] = from; cut[i] = from; continue; }   // swallowed whole
        cut[i] = from;
        stop[i] = parse_chunk(from, cut[i + 1], e, flavour, chunks[i]);
    }
    rth_scene* s = new rth_scene();
    const int rc = stitch(chunks, flavour, s);
    if (rc != RT_OK) { delete s; return rc; }
    *out = s;
    return RT_OK;
}

const char kMagic[8] = {'R', 'T', 'S', 'C', '0', '0', '0', '1'};
struct CacheHeader {
    char magic[8];
    int32_t flavour, resx, resy, nlight;
    float eye[3], view[3], fov, pad;
    int64_t nprim, nmaterial, nobject;
};

}  // namespace

extern "C" int rth_scene_parse_text(const char* text, int64_t len, int flavour, int nthreads, rth_scene** out) {
    if (!text || len < 0 || !out) return RT_E_INVALID;
    return parse_range(text, text + len, flavour, nthreads, out);
}

extern "C" int rth_scene_parse_file(const char* path, int flavour, int nthreads, rth_scene** out) {
    if (!path || !out) return RT_E_INVALID;
    const int fd = open(path, O_RDONLY);
    if (fd < 0) return RT_E_INVALID;
    struct stat sb;
    if (fstat(fd, &sb) != 0) { close(fd); return RT_E_INVALID; }
    if (sb.st_size == 0) { close(fd); return parse_range("", "", flavour, 1, out); }
    void* m = mmap(nullptr, (size_t)sb.st_size, PROT_READ, MAP_PRIVATE, fd, 0);
    close(fd);
    if (m == MAP_FAILED) return RT_E_INVALID;
    const int rc = parse_range((const char*)m, (const char*)m + sb.st_size, flavour, nthreads, out);
    munmap(m, (size_t)sb.st_size);
    return rc;
}

extern "C" void rth_scene_free(rth_scene* s) { delete s; }

extern "C" int rth_scene_fill_desc(const rth_scene* s, rt_scene_desc* d) {
    if (!s || !d) return RT_E_INVALID;
    memset(d, 0, sizeof *d);
    const bool fin = s->flavour == RTH_FLAVOUR_FINAL;
    d->abi_version = RT_ABI_VERSION;
    d->flags = RT_SCENE_KD_REFERENCE | (fin ? RT_SCENE_CULLING : 0u);        // hw4.cpp:40 / parameter.h:13
    d->nprim = (int64_t)s->type.size();
    d->prims = reinterpret_cast<const rt_prim*>(s->prims.data());
    d->prim_type = s->type.data();
    d->prim_material = s->mat.data();
    d->nmaterial = (int32_t)(s->materials.size() / 10);
    d->materials = reinterpret_cast<const rt_material*>(s->materials.data());
    d->nobject = (int32_t)s->obj_first.size();
    d->obj_first = s->obj_first.data(); d->obj_count = s->obj_count.data();
    d->nlight = (int32_t)s->lights.size();
    d->lights = s->lights.data();
    d->tolerance_depth = fin ? 0.001 : 0.00001;                                // parameter.h:9 / hw4.cpp:19
    d->shadow_bias = 0.0001f; d->refract_bias = 0.01f;                         // hw4.cpp:34-35
    d->device = -1;
    return RT_OK;
}

extern "C" int rth_scene_view(const rth_scene* s, float eye[3], float view[3], float* fov, int32_t* resx, int32_t* resy) {
    if (!s) return RT_E_INVALID;
    if (eye) memcpy(eye, s->eye, sizeof s->eye);
    if (view) memcpy(view, s->view, sizeof s->view);
    if (fov) *fov = s->fov;
    if (resx) *resx = s->resx;
    if (resy) *resy = s->resy;
    return RT_OK;
}

extern "C" int rth_scene_save(const rth_scene* s, const char* path) {
    if (!s || !path) return RT_E_INVALID;
    FILE* f = fopen(path, "wb");
    if (!f) return RT_E_INVALID;
    CacheHeader h; memset(&h, 0, sizeof h);
    memcpy(h.magic, kMagic, 8);
    h.flavour = s->flavour; h.resx = s->resx; h.resy = s->resy; h.nlight = (int32_t)s->lights.size();
    memcpy(h.eye, s->eye, sizeof h.eye); memcpy(h.view, s->view, sizeof h.view); h.fov = s->fov;
    h.nprim = (int64_t)s->type.size(); h.nmaterial = (int64_t)(s->materials.size() / 10); h.nobject = (int64_t)s->obj_first.size();
    bool ok = fwrite(&h, sizeof h, 1, f) == 1;
    auto put = [&](const void* p, size_t bytes) { if (bytes) ok = ok && fwrite(p, 1, bytes, f) == bytes; };
    put(s->prims.data(), s->prims.size() * 4); put(s->type.data(), s->type.size()); put(s->mat.data(), s->mat.size() * 4);
    put(s->materials.data(), s->materials.size() * 4);
    put(s->obj_first.data(), s->obj_first.size() * 8); put(s->obj_count.data(), s->obj_count.size() * 8);
    put(s->lights.data(), s->lights.size() * sizeof(rt_light));
    ok = (fclose(f) == 0) && ok;
    return ok ? RT_OK : RT_E_INVALID;
}

extern "C" int rth_scene_load(const char* path, rth_scene** out) {
    if (!path || !out) return RT_E_INVALID;
    FILE* f = fopen(path, "rb");
    if (!f) return RT_E_INVALID;
    CacheHeader h;
    if (fread(&h, sizeof h, 1, f) != 1 || memcmp(h.magic, kMagic, 8) != 0 || h.nprim < 0 || h.nmaterial < 0 ||
        h.nobject < 0 || h.nlight < 0 || (h.flavour != RTH_FLAVOUR_HW4 && h.flavour != RTH_FLAVOUR_FINAL)) { fclose(f); return RT_E_INVALID; }
    struct stat sb;
    const int64_t need = (int64_t)sizeof h + h.nprim * 53 + h.nmaterial * 40 + h.nobject * 16 + (int64_t)h.nlight * (int64_t)sizeof(rt_light);
    if (fstat(fileno(f), &sb) != 0 || (int64_t)sb.st_size != need) { fclose(f); return RT_E_INVALID; }
    rth_scene* s = new rth_scene();
    s->flavour = h.flavour; s->resx = h.resx; s->resy = h.resy; s->fov = h.fov;
    memcpy(s->eye, h.eye, sizeof h.eye); memcpy(s->view, h.view, sizeof h.view);
    s->prims.resize((size_t)h.nprim * 12); s->type.resize((size_t)h.nprim); s->mat.resize((size_t)h.nprim);
    s->materials.resize((size_t)h.nmaterial * 10);
    s->obj_first.resize((size_t)h.nobject); s->obj_count.resize((size_t)h.nobject);
    s->lights.resize((size_t)h.nlight);
    bool ok = true;
    auto get = [&](void* p, size_t bytes) { if (bytes) ok = ok && fread(p, 1, bytes, f) == bytes; };
    get(s->prims.data(), s->prims.size() * 4); get(s->type.data(), s->type.size()); get(s->mat.data(), s->mat.size() * 4);
    get(s->materials.data(), s->materials.size() * 4);
    get(s->obj_first.data(), s->obj_first.size() * 8); get(s->obj_count.data(), s->obj_count.size() * 8);
    get(s->lights.data(), s->lights.size() * sizeof(rt_light));
    fclose(f);
    for (int32_t m : s->mat) if (m < 0 || m >= h.nmaterial) { ok = false; break; }
    for (size_t i = 0; ok && i < s->obj_first.size(); ++i)
        if (s->obj_first[i] < 0 || s->obj_count[i] < 0 || s->obj_first[i] + s->obj_count[i] > h.nprim) ok = false;
    if (!ok) { delete s; return RT_E_INVALID; }
    *out = s;
    return RT_OK;
}
