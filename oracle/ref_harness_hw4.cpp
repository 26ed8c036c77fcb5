/*
 * ref_harness_hw4.cpp -- TEST INFRASTRUCTURE ONLY.
 *
 * Thin extern "C" wrappers that call the UNMODIFIED reference functions of
 * HW4/hw4/hw4.cpp.  The reference translation unit is pulled in by path at
 * compile time (oracle/Makefile passes -I/root/reference/HW4/hw4); no reference
 * source is copied into this repository.  The result, oracle/_ref/libref_hw4.so,
 * is git-ignored; it is used to pin oracle/rt_oracle.c and to generate the
 * fixtures under tests/golden/ (oracle/gen_golden.py).
 */
#include <cstdint>
#define main ref_main_hw4
#include "hw4.cpp"          /* found through -I, never copied */
#undef main

namespace {
struct Quiet {
    std::streambuf* old;
    Quiet() : old(std::cout.rdbuf(nullptr)) {}
    ~Quiet() { std::cout.rdbuf(old); }
};

int flat_count(KDNode* n, int64_t* leafrefs) {
    int c = 1;
    if (n->isLeaf) { *leafrefs += (int64_t)n->kdMeshes.size(); return c; }
    c += flat_count(n->leftChild, leafrefs);
    c += flat_count(n->rightChild, leafrefs);
    return c;
}

std::vector<SimpleMesh*> g_index;   /* prim id -> mesh */
int prim_id(SimpleMesh* m) {
    for (size_t i = 0; i < g_index.size(); i++) if (g_index[i] == m) return (int)i;
    return -1;
}

int32_t flat_rec(KDNode* n, float* lo, float* hi, int32_t* left, int32_t* right, int32_t* is_leaf,
                 int64_t* leaf_first, int32_t* leaf_count, int32_t* leaf_prims, int64_t* nn, int64_t* nl) {
    int32_t me = (int32_t)(*nn)++;
    for (int k = 0; k < 3; k++) {
        lo[3 * me + k] = n->center[k] - n->extend[k] / 2.0f;
        hi[3 * me + k] = n->center[k] + n->extend[k] / 2.0f;
    }
    is_leaf[me] = n->isLeaf; left[me] = right[me] = -1; leaf_first[me] = 0; leaf_count[me] = 0;
    if (n->isLeaf) {
        leaf_first[me] = *nl; leaf_count[me] = (int32_t)n->kdMeshes.size();
        for (size_t i = 0; i < n->kdMeshes.size(); i++) leaf_prims[(*nl)++] = prim_id(n->kdMeshes[i]);
    } else {
        left[me] = flat_rec(n->leftChild, lo, hi, left, right, is_leaf, leaf_first, leaf_count, leaf_prims, nn, nl);
        right[me] = flat_rec(n->rightChild, lo, hi, left, right, is_leaf, leaf_first, leaf_count, leaf_prims, nn, nl);
    }
    return me;
}
}  // namespace

extern "C" {

void ref_reset() {
    for (size_t i = 0; i < simpleMeshes.size(); i++) delete simpleMeshes[i];
    for (size_t i = 0; i < simpleObjects.size(); i++) { simpleObjects[i]->kdRoot->Clean(); delete simpleObjects[i]->kdRoot; delete simpleObjects[i]; }
    for (size_t i = 0; i < simpleLights.size(); i++) delete simpleLights[i];
    simpleMeshes.clear(); simpleObjects.clear(); simpleLights.clear(); g_index.clear();
    builded = 0; maxDepth = 0; KDNode::index = 0; rayHitAnything = false;
}

/* prims: [n][12] (tri: 9 vertex floats + normal[0]; sphere: centre, radius)
 * mats: [nmat][10] in `M` line order.  Objects are built exactly as the `O`
 * handler does (hw4.cpp:729-746).  lights: [nl][7] pos, colour, intensity. */
void ref_scene_build(int64_t n, const float* prims, const uint8_t* types, const int32_t* matidx,
                     const float* mats, int32_t nobj, const int64_t* first, const int64_t* count,
                     int32_t nl, const float* lights) {
    Quiet q;
    ref_reset();
    for (int64_t i = 0; i < n; i++) {
        const float* f = prims + 12 * i; const float* m = mats + 10 * matidx[i];
        SimpleMesh* mesh;
        if (types[i] == 1) {
            Sphere* s = new Sphere(); s->center = vec3(f[0], f[1], f[2]); s->radius = f[3]; mesh = s;
        } else {
            Triangle* t = new Triangle();
            for (int j = 0; j < 3; j++) t->vertices[j] = vec3(f[3 * j], f[3 * j + 1], f[3 * j + 2]);
            t->normal[0] = t->normal[1] = t->normal[2] = vec3(f[9], f[10], f[11]);
            mesh = t;
        }
        mesh->surfColor = vec3(m[0], m[1], m[2]); mesh->Ka = m[3]; mesh->Kd = m[4]; mesh->Ks = m[5];
        mesh->specularity = m[6]; mesh->reflect = m[7]; mesh->refract = m[8]; mesh->Nr = m[9];
        simpleMeshes.push_back(mesh); g_index.push_back(mesh);
    }
    for (int32_t o = 0; o < nobj; o++) {
        KDNode* kd = new KDNode();
        for (int64_t i = first[o]; i < first[o] + count[o]; i++) kd->kdMeshes.push_back(simpleMeshes[i]);
        if (kd->kdMeshes.size() == 0) { delete kd; continue; }
        float tempF = log10((int)kd->kdMeshes.size()) - 1.0f;
        kd->modelScale = tempF < 1.5f ? 1.5f : tempF;
        kd->BuildKD(0);
        SimpleObject* obj = new SimpleObject(); obj->kdRoot = kd; simpleObjects.push_back(obj);
    }
    for (int32_t i = 0; i < nl; i++) {
        PointLight* pl = new PointLight(); const float* L = lights + 7 * i;
        pl->pos = vec3(L[0], L[1], L[2]); pl->color = vec3(L[3], L[4], L[5]); pl->intensity = L[6];
        simpleLights.push_back(pl);
    }
}

void ref_kd_stats(int64_t* nbuilt, int32_t* maxdepth, int64_t* flat_nodes, int64_t* leaf_refs) {
    *nbuilt = builded; *maxdepth = maxDepth; *flat_nodes = 0; *leaf_refs = 0;
    for (size_t i = 0; i < simpleObjects.size(); i++) *flat_nodes += flat_count(simpleObjects[i]->kdRoot, leaf_refs);
}

void ref_kd_dump(float* lo, float* hi, int32_t* left, int32_t* right, int32_t* is_leaf,
                 int64_t* leaf_first, int32_t* leaf_count, int32_t* leaf_prims, int32_t* roots) {
    int64_t nn = 0, nl = 0;
    for (size_t i = 0; i < simpleObjects.size(); i++)
        roots[i] = flat_rec(simpleObjects[i]->kdRoot, lo, hi, left, right, is_leaf, leaf_first, leaf_count, leaf_prims, &nn, &nl);
}

int ref_box_test(const float* c, const float* e, const float* o, const float* d) {
    KDNode n; n.center = vec3(c[0], c[1], c[2]); n.extend = vec3(e[0], e[1], e[2]);
    return n.HitTest(vec3(o[0], o[1], o[2]), vec3(d[0], d[1], d[2])) ? 1 : 0;
}

int ref_tri_test(const float* v, const float* o, const float* d, float* point) {
    Triangle t(vec3(v[0], v[1], v[2]), vec3(v[3], v[4], v[5]), vec3(v[6], v[7], v[8]));
    bool b = t.IntersectionTest(vec3(d[0], d[1], d[2]), vec3(o[0], o[1], o[2]));
    if (b) { point[0] = t.tempIntersectionPoint[0]; point[1] = t.tempIntersectionPoint[1]; point[2] = t.tempIntersectionPoint[2]; }
    return b ? 1 : 0;
}

int ref_sphere_test(const float* c, float r, const float* o, const float* d, float* point) {
    Sphere s(vec3(c[0], c[1], c[2]), r);
    bool b = s.IntersectionTest(vec3(d[0], d[1], d[2]), vec3(o[0], o[1], o[2]));
    if (b) { point[0] = s.tempIntersectionPoint[0]; point[1] = s.tempIntersectionPoint[1]; point[2] = s.tempIntersectionPoint[2]; }
    return b ? 1 : 0;
}

/* SimpleMeshRayTrace itself (hw4.cpp:794); returns its colour and the
 * rayHitAnything flag.  depth as given (0 = primary). */
int ref_trace_colour(const float* o, const float* d, int depth, const float* eye, float nr, float* rgb) {
    Camera cam; cam.center = vec3(eye[0], eye[1], eye[2]);
    rayHitAnything = false;
    vec3 c = SimpleMeshRayTrace(vec3(o[0], o[1], o[2]), vec3(d[0], d[1], d[2]), depth, cam, nr);
    rgb[0] = c[0]; rgb[1] = c[1]; rgb[2] = c[2];
    int h = rayHitAnything ? 1 : 0; rayHitAnything = false;
    return h;
}

/* Closest hit through the unmodified SimpleMeshRayTrace: the caller built the
 * scene with ID-colour materials (surfColor.x = id+1, Ka = 1, everything else
 * 0) and no lights, so the returned colour IS the hit id, and the hit mesh's
 * tempIntersectionPoint still holds the hit point (hw4.cpp:850-857). */
void ref_trace_ids(const float* rays /* [n][6] */, int64_t n, int32_t* prim, float* len, float* point) {
    Camera cam; cam.center = vec3(0, 0, 0);
    for (int64_t i = 0; i < n; i++) {
        const float* r = rays + 6 * i;
        vec3 ori(r[0], r[1], r[2]), dir(r[3], r[4], r[5]);
        vec3 c = SimpleMeshRayTrace(ori, dir, 0, cam, 1.0f);
        int id = (int)c[0] - 1;
        prim[i] = id;
        if (id >= 0) {
            vec3 v = simpleMeshes[id]->tempIntersectionPoint;
            len[i] = (ori - v).length();
            point[3 * i] = v[0]; point[3 * i + 1] = v[1]; point[3 * i + 2] = v[2];
        } else { len[i] = 1e9f; point[3 * i] = point[3 * i + 1] = point[3 * i + 2] = 0; }
    }
    rayHitAnything = false;
}

/* IsInShadow itself (hw4.cpp:957) */
int ref_in_shadow(const float* point, const float* light, const float* normal) {
    return IsInShadow(vec3(point[0], point[1], point[2]), vec3(light[0], light[1], light[2]), 0,
                      vec3(normal[0], normal[1], normal[2])) ? 1 : 0;
}

/* Camera::CalcPlaneParameters + GetPixelCenter + normalize (hw4.cpp:242-276,1053-1056) */
void ref_camera(const float* eye, const float* view_raw, float fov, int resx, int resy, float* out /* center3, corners12, x3, y3 */) {
    Quiet q;
    Camera cam; cam.planeDistance = PLANE_DISTANCE; cam.up = upVec;
    cam.center = vec3(eye[0], eye[1], eye[2]);
    cam.viewDirection = vec3(view_raw[0], view_raw[1], view_raw[2]);
    cam.viewDirection = cam.viewDirection.normalize();
    cam.fovX = fov; cam.fovY = fov / ASPECT_RATIO; cam.resX = resx; cam.resY = resy;
    cam.CalcPlaneParameters();
    for (int k = 0; k < 3; k++) {
        out[k] = cam.center[k];
        for (int c = 0; c < 4; c++) out[3 + 3 * c + k] = cam.planeCorners[c][k];
        out[15 + k] = cam.planeXAxis[k]; out[18 + k] = cam.planeYAxis[k];
    }
}

void ref_primary_rays(const float* camv /* as ref_camera wrote */, int resx, int resy,
                      const float* xy /* [n][2] */, int64_t n, float* rays /* [n][6] */) {
    Camera cam;
    for (int k = 0; k < 3; k++) {
        cam.center[k] = camv[k];
        for (int c = 0; c < 4; c++) cam.planeCorners[c][k] = camv[3 + 3 * c + k];
        cam.planeXAxis[k] = camv[15 + k]; cam.planeYAxis[k] = camv[18 + k];
    }
    cam.resX = resx; cam.resY = resy;
    for (int64_t i = 0; i < n; i++) {
        vec2 vv; vv[0] = xy[2 * i]; vv[1] = xy[2 * i + 1];
        vec3 pc = cam.GetPixelCenter(vv);
        vec3 ray = (pc - cam.center).normalize();
        for (int k = 0; k < 3; k++) { rays[6 * i + k] = cam.center[k]; rays[6 * i + 3 + k] = ray[k]; }
    }
}

}  // extern "C"
