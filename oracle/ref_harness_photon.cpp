/*
 * ref_harness_photon.cpp -- TEST INFRASTRUCTURE ONLY.
 *
 * extern "C" wrappers around the UNMODIFIED PhotonMap of Final/photon/photon.cpp
 * (compiled from /root/reference by oracle/Makefile and linked here; nothing
 * is copied).  Used to pin the photon part of oracle/rt_oracle.c and to
 * generate tests/golden/photon_*.npz.
 */
#include <cstdint>
#include <cstring>
#include <iostream>
#include "photon.h"         /* found through -I/root/reference/Final/photon */

static PhotonMap* g_map = nullptr;

namespace {
struct Quiet {
    std::streambuf* old;
    Quiet() : old(std::cout.rdbuf(nullptr)) {}
    ~Quiet() { std::cout.rdbuf(old); }
};
}

extern "C" {

void refp_create(int max_photons) {
    Quiet q;
    delete g_map;
    g_map = new PhotonMap(max_photons);
    InitDirectionTable();
}

void refp_tables(float* cost, float* sint, float* cosp, float* sinp) {
    InitDirectionTable();
    memcpy(cost, cosThetaTable, 1024); memcpy(sint, sinThetaTable, 1024);
    memcpy(cosp, cosPhiTable, 1024); memcpy(sinp, sinPhiTable, 1024);
}

/* AddPhoton (photon.cpp:835) for n photons */
void refp_add(const float* pos, const float* dir, const float* power, int64_t n) {
    for (int64_t i = 0; i < n; i++)
        g_map->AddPhoton(vec3(pos[3 * i], pos[3 * i + 1], pos[3 * i + 2]),
                         vec3(dir[3 * i], dir[3 * i + 1], dir[3 * i + 2]),
                         vec3(power[3 * i], power[3 * i + 1], power[3 * i + 2]));
}

int refp_stored() { return g_map->storedPhotonNum; }
int refp_record_size() { return (int)sizeof(Photon); }

void refp_balance() { g_map->balance(); }
void refp_scale(float s) { g_map->ScalePhotonPower(s, 0, g_map->storedPhotonNum); }

/* raw copy of photons[0..stored] (40-byte records) */
void refp_dump(void* out) { memcpy(out, g_map->photons, sizeof(Photon) * (size_t)(g_map->storedPhotonNum + 1)); }
/* overwrite the stored array with a caller-provided one (already balanced or not) */
void refp_load(const void* in, int n) {
    memcpy(g_map->photons, in, sizeof(Photon) * (size_t)(n + 1));
    g_map->storedPhotonNum = n;
    g_map->half_stored_photons = n / 2 - 1;
}

/* LocatePhoton (photon.cpp:53) driven as CalcIrradiance drives it (:397-407) */
int refp_locate(const float* pos, int k, float maxDist, int32_t* idx, float* dist2, float* radius2) {
    NearestPhotons np;
    np.dist2 = (float*)malloc(sizeof(float) * (k + 1));
    np.index = (Photon**)malloc(sizeof(Photon*) * (k + 1));
    np.pos[0] = pos[0]; np.pos[1] = pos[1]; np.pos[2] = pos[2];
    np._max = k; np.found = 0; np.got_heap = 0;
    np.dist2[0] = maxDist * maxDist;
    g_map->LocatePhoton(&np, 1);
    for (int i = 0; i < k; i++) {
        idx[i] = i < np.found ? (int32_t)(np.index[i + 1] - g_map->photons) : -1;
        dist2[i] = i < np.found ? np.dist2[i + 1] : 0.0f;
    }
    *radius2 = np.dist2[0];
    int f = np.found;
    free(np.dist2); free(np.index);
    return f;
}

/* CalcIrradiance (photon.cpp:342) */
void refp_irradiance(const float* pos, const float* nrm, int64_t q, float maxDist, int k, float* rgb) {
    for (int64_t i = 0; i < q; i++) {
        vec3 c = g_map->CalcIrradiance(vec3(pos[3 * i], pos[3 * i + 1], pos[3 * i + 2]),
                                       vec3(nrm[3 * i], nrm[3 * i + 1], nrm[3 * i + 2]), maxDist, k, 8);
        rgb[3 * i] = c[0]; rgb[3 * i + 1] = c[1]; rgb[3 * i + 2] = c[2];
    }
}

}  // extern "C"
