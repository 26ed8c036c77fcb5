/*
 * ref_harness_final.cpp -- TEST INFRASTRUCTURE ONLY.
 *
 * extern "C" wrappers around the UNMODIFIED functions of Final/photon/final.cpp and photon.cpp
 * (SimpleMeshRayTrace, IsInShadow, TestVisualizeLight, RectLight::GetSamplePoses, TraceIrradiance,
 * PhotonTrace, PhotonTraceCaustics, AddPhoton, balance, mRandom).  The reference translation unit is
 * pulled in by path at compile time (oracle/Makefile: -I/root/reference/Final/photon); nothing is
 * copied.  Scene objects are created the way the reference's parser creates them (final.cpp:575-733).
 * Result: oracle/_ref/libref_final.so (git-ignored), used by oracle/gen_golden.py only.
 */
#include <cstdint>
#include <cstring>
#define main ref_main_final
#include "final.cpp"        /* found through -I, never copied */
#undef main

/* Observation of the emission loops (final.cpp:1053-1123).  They are inline in PhotonMappingImageSynthesis and hand their
 * rays straight to PhotonTrace / PhotonTraceCaustics (photon.cpp, another translation unit).  The library is linked with
 * -Wl,--wrap=<the two mangled names> (oracle/Makefile), so THIS file's calls to them arrive at the two functions below,
 * which note the depth-0 arguments and call the real, unmodified functions (__real_...); photon.cpp's own recursive
 * calls are not affected.  No reference file is edited or replaced. */
#define PT_SYM  "_Z11PhotonTraceR9PhotonMapRSt6vectorIP10SimpleMeshSaIS3_EE4vec3S7_ifS7_"
#define PTC_SYM "_Z19PhotonTraceCausticsR9PhotonMapRSt6vectorIP10SimpleMeshSaIS3_EE4vec3S7_ifS7_bS6_"
void real_PhotonTrace(PhotonMap&, vector<SimpleMesh*>&, vec3, vec3, int, float, vec3) asm("__real_" PT_SYM);
void wrap_PhotonTrace(PhotonMap&, vector<SimpleMesh*>&, vec3, vec3, int, float, vec3) asm("__wrap_" PT_SYM);
void real_PhotonTraceCaustics(PhotonMap&, vector<SimpleMesh*>&, vec3, vec3, int, float, vec3, bool, vector<SimpleMesh*>&) asm("__real_" PTC_SYM);
void wrap_PhotonTraceCaustics(PhotonMap&, vector<SimpleMesh*>&, vec3, vec3, int, float, vec3, bool, vector<SimpleMesh*>&) asm("__wrap_" PTC_SYM);
static std::vector<float> g_emit_log[2];        /* per loop: pos xyz, dir xyz of every emission */
static std::vector<int> g_emit_stored[2];       /* photons stored in the map when the emission was made */
static bool g_emit_record = false;
void wrap_PhotonTrace(PhotonMap& m, vector<SimpleMesh*>& v, vec3 ori, vec3 dir, int depth, float nr, vec3 col) {
    if (g_emit_record && depth == 0) {
        for (int k = 0; k < 3; k++) g_emit_log[0].push_back(ori[k]);
        for (int k = 0; k < 3; k++) g_emit_log[0].push_back(dir[k]);
        g_emit_stored[0].push_back(m.storedPhotonNum);
    }
    real_PhotonTrace(m, v, ori, dir, depth, nr, col);
}
void wrap_PhotonTraceCaustics(PhotonMap& m, vector<SimpleMesh*>& v, vec3 ori, vec3 dir, int depth, float nr, vec3 col, bool flag,
                              vector<SimpleMesh*>& cm) {
    if (g_emit_record && depth == 0) {
        for (int k = 0; k < 3; k++) g_emit_log[1].push_back(ori[k]);
        for (int k = 0; k < 3; k++) g_emit_log[1].push_back(dir[k]);
        g_emit_stored[1].push_back(m.storedPhotonNum);
    }
    real_PhotonTraceCaustics(m, v, ori, dir, depth, nr, col, flag, cm);
}

namespace {
struct Quiet {
    std::streambuf* old;
    Quiet() : old(std::cout.rdbuf(nullptr)) {}
    ~Quiet() { std::cout.rdbuf(old); }
};
PhotonMap* g_maps[2] = {nullptr, nullptr};
void set_material(SimpleMesh* m, const float* a) {
    m->surfColor = vec3(a[0], a[1], a[2]); m->Ka = a[3]; m->Kd = a[4]; m->Ks = a[5];
    m->specularity = a[6]; m->reflect = a[7]; m->refract = a[8]; m->Nr = a[9];
}
vec3 V3(const float* p) { return vec3(p[0], p[1], p[2]); }
}

extern "C" {

int reff_seed() { return RT_REF_FIXED_SEED; }

void reff_clear() {
    for (size_t i = 0; i < simpleMeshes.size(); i++) delete simpleMeshes[i];
    for (size_t i = 0; i < lightMeshes.size(); i++) delete lightMeshes[i];
    for (size_t i = 0; i < causticMeshes.size(); i++) delete causticMeshes[i];
    for (size_t i = 0; i < simpleLights.size(); i++) delete simpleLights[i];
    simpleMeshes.clear(); lightMeshes.clear(); causticMeshes.clear(); simpleLights.clear();
    rayHitAnything = false;
    InitDirectionTable();
}

/* `S` line, final.cpp:540-573 */
void reff_add_sphere(const float* c4, const float* mat) {
    Sphere* s = new Sphere();
    s->center = vec3(c4[0], c4[1], c4[2]); s->radius = c4[3];
    set_material(s, mat);
    simpleMeshes.push_back(s);
}

/* `T` line, final.cpp:575-623 */
void reff_add_triangle(const float* v9, const float* mat) {
    Quiet q;
    Triangle* t = new Triangle();
    t->vertices[0] = V3(v9); t->vertices[1] = V3(v9 + 3); t->vertices[2] = V3(v9 + 6);
    set_material(t, mat);
    vec3 tempCross = (t->vertices[2] - t->vertices[0]) ^ (t->vertices[1] - t->vertices[0]);
    t->normal[0] = tempCross.normalize();
    t->normal[1] = tempCross.normalize();
    t->normal[2] = tempCross.normalize();
    simpleMeshes.push_back(t);
}

/* `RectL` line, final.cpp:683-733 */
void reff_add_rectlight(const float* pos, float size1, float size2, int n1, int n2) {
    RectLight* rl = new RectLight();
    rl->pos = V3(pos); rl->dir1 = vec3(1.0, 0, 0); rl->dir2 = vec3(0, 0, 1.0);
    rl->size1 = size1; rl->size2 = size2; rl->sampleCountDir1 = n1; rl->sampleCountDir2 = n2;
    simpleLights.push_back(rl);
    Triangle* triangle = new Triangle();
    triangle->vertices[0] = rl->pos - rl->dir1 * rl->size1 * 0.5 - rl->dir2 * rl->size2 * 0.5;
    triangle->vertices[1] = rl->pos + rl->dir1 * rl->size1 * 0.5 - rl->dir2 * rl->size2 * 0.5;
    triangle->vertices[2] = rl->pos - rl->dir1 * rl->size1 * 0.5 + rl->dir2 * rl->size2 * 0.5;
    vec3 tempCross = (triangle->vertices[2] - triangle->vertices[0]) ^ (triangle->vertices[1] - triangle->vertices[0]);
    triangle->normal[0] = tempCross.normalize(); triangle->normal[1] = tempCross.normalize(); triangle->normal[2] = tempCross.normalize();
    lightMeshes.push_back(triangle);
    triangle = new Triangle();
    triangle->vertices[0] = rl->pos + rl->dir1 * rl->size1 * 0.5 + rl->dir2 * rl->size2 * 0.5;
    triangle->vertices[1] = rl->pos + rl->dir1 * rl->size1 * 0.5 - rl->dir2 * rl->size2 * 0.5;
    triangle->vertices[2] = rl->pos - rl->dir1 * rl->size1 * 0.5 + rl->dir2 * rl->size2 * 0.5;
    tempCross = (triangle->vertices[2] - triangle->vertices[0]) ^ (triangle->vertices[1] - triangle->vertices[0]);
    triangle->normal[0] = tempCross.normalize(); triangle->normal[1] = tempCross.normalize(); triangle->normal[2] = tempCross.normalize();
    lightMeshes.push_back(triangle);
}

/* the hard-coded proxy of final.cpp:754-758 */
void reff_add_caustic_sphere(const float* c4) {
    Sphere* s = new Sphere();
    s->center = vec3(c4[0], c4[1], c4[2]); s->radius = c4[3];
    causticMeshes.push_back(s);
}

void reff_camera_center(const float* eye) { camera.center = V3(eye); }

void reff_light_quads(float* out /* [nquads*2][9] */) {
    for (size_t i = 0; i < lightMeshes.size(); i++) {
        Triangle* t = (Triangle*)lightMeshes[i];
        for (int v = 0; v < 3; v++) for (int k = 0; k < 3; k++) out[9 * i + 3 * v + k] = t->vertices[v][k];
    }
}

int reff_light_samples(int light, float* out3, int cap) {
    vector<vec3> s = simpleLights[light]->GetSamplePoses();
    for (size_t i = 0; i < s.size() && (int)i < cap; i++) { out3[3 * i] = s[i][0]; out3[3 * i + 1] = s[i][1]; out3[3 * i + 2] = s[i][2]; }
    return (int)s.size();
}

void reff_is_in_shadow(const float* point, const float* light, const float* normal, int64_t n, uint8_t* out) {
    for (int64_t i = 0; i < n; i++)
        out[i] = IsInShadow(V3(point + 3 * i), V3(light + 3 * i), nullptr, V3(normal + 3 * i)) ? 1 : 0;
}

void reff_sees_light(const float* ori, const float* dir, int64_t n, uint8_t* out) {
    for (int64_t i = 0; i < n; i++) out[i] = TestVisualizeLight(V3(ori + 3 * i), V3(dir + 3 * i)) ? 1 : 0;
}

void reff_maps_create(int max_global, int max_caustic) {
    Quiet q;
    delete g_maps[0]; delete g_maps[1];
    g_maps[0] = new PhotonMap(max_global); g_maps[1] = new PhotonMap(max_caustic);
}
int reff_map_stored(int which) { return g_maps[which]->storedPhotonNum; }
void reff_map_dump(int which, void* out) { memcpy(out, g_maps[which]->photons, sizeof(Photon) * (size_t)(g_maps[which]->storedPhotonNum + 1)); }
void reff_map_load(int which, const void* in, int n) {
    memcpy(g_maps[which]->photons, in, sizeof(Photon) * (size_t)(n + 1));
    g_maps[which]->storedPhotonNum = n; g_maps[which]->half_stored_photons = n / 2 - 1;
}
void reff_map_balance_scale(int which) {       /* final.cpp:1125-1129 */
    PhotonMap& m = *g_maps[which];
    m.balance();
    m.ScalePhotonPower(1.0f / ((float)m.storedPhotonNum >= 1.0 ? (float)m.storedPhotonNum : 1.0), 0, m.storedPhotonNum);
}

/* draws of final.cpp's copy of mRandom (photon.h:16-18) */
void reff_random(int n, float* out) { for (int i = 0; i < n; i++) out[i] = mRandom(); }

/* PhotonTrace / PhotonTraceCaustics with the caller's emission position, direction, colour */
void reff_photon_trace(int which, const float* pos, const float* dir, const float* color, float nr, int64_t n) {
    for (int64_t i = 0; i < n; i++) {
        if (which == 0) PhotonTrace(*g_maps[0], simpleMeshes, V3(pos + 3 * i), V3(dir + 3 * i), 0, nr, V3(color));
        else PhotonTraceCaustics(*g_maps[1], simpleMeshes, V3(pos + 3 * i), V3(dir + 3 * i), 0, nr, V3(color), false, causticMeshes);
    }
}

void reff_trace_irradiance(const float* ori, const float* dir, int depth, float nr, int64_t n, float* rgb) {
    for (int64_t i = 0; i < n; i++) {
        vec3 c = TraceIrradiance(*g_maps[0], simpleMeshes, V3(ori + 3 * i), V3(dir + 3 * i), depth, nr);
        rgb[3 * i] = c[0]; rgb[3 * i + 1] = c[1]; rgb[3 * i + 2] = c[2];
    }
}

/* SimpleMeshRayTrace(pmap, cpmap, ...), final.cpp:785-983 */
void reff_eye_trace(const float* ori, const float* dir, int depth, float nr, int64_t n, float* rgb, uint8_t* hit_any) {
    for (int64_t i = 0; i < n; i++) {
        rayHitAnything = false;
        vec3 c = SimpleMeshRayTrace(*g_maps[0], *g_maps[1], V3(ori + 3 * i), V3(dir + 3 * i), depth, camera, nr);
        rgb[3 * i] = c[0]; rgb[3 * i + 1] = c[1]; rgb[3 * i + 2] = c[2];
        hit_any[i] = rayHitAnything ? 1 : 0;
    }
}

/* The reference's own emission loops, run once: PhotonMappingImageSynthesis (final.cpp:1045-1209) with a 0 x 0 camera,
 * so that its pixel loop has no iteration (it still writes an empty colorOutput.ppm into the current directory: call
 * this from a scratch directory).  Draws come from final.cpp's copy of the serial engine, which must be fresh (nothing
 * else of this library called before in the process) for the fixture to start at the seed.  Returns the emission counts. */
void reff_run_emission_loops(int64_t* n_global, int64_t* n_caustic) {
    Quiet q;
    g_emit_log[0].clear(); g_emit_log[1].clear(); g_emit_stored[0].clear(); g_emit_stored[1].clear();
    g_emit_record = true;
    Camera cam = camera;
    cam.resX = 0; cam.resY = 0;
    PhotonMappingImageSynthesis(cam);
    g_emit_record = false;
    *n_global = (int64_t)g_emit_stored[0].size(); *n_caustic = (int64_t)g_emit_stored[1].size();
}
void reff_emission_log(int which, float* posdir6, int32_t* stored_before) {
    memcpy(posdir6, g_emit_log[which].data(), g_emit_log[which].size() * sizeof(float));
    memcpy(stored_before, g_emit_stored[which].data(), g_emit_stored[which].size() * sizeof(int));
}

}  // extern "C"
