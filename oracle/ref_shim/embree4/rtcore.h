// Stand-in for the subset of Embree 4's public C API (third-party) that the reference's
// embree_sycl kernel uses: the ray / hit records, the query-argument structs and the two
// single-ray queries. TEST INFRASTRUCTURE (oracle/Makefile, target _ref). The queries are
// implemented in oracle/ref_driver.cpp by a brute-force intersector with the semantics DESIGN.md
// documents for the Embree stand-in; everything else here is declarations.
#pragma once
#include <cstddef>
#include <cstdint>

#define RTC_INVALID_GEOMETRY_ID ((unsigned int)-1)
#define RTC_MAX_INSTANCE_LEVEL_COUNT 1

typedef struct RTCDeviceTy *RTCDevice;
typedef struct RTCSceneTy *RTCScene;
typedef struct RTCGeometryTy *RTCGeometry;

struct RTCRay {
    float org_x, org_y, org_z, tnear;
    float dir_x, dir_y, dir_z, time;
    float tfar;
    unsigned int mask, id, flags;
};
struct RTCHit {
    float Ng_x, Ng_y, Ng_z;
    float u, v;
    unsigned int primID, geomID;
    unsigned int instID[RTC_MAX_INSTANCE_LEVEL_COUNT];
};
struct RTCRayHit {
    RTCRay ray;
    RTCHit hit;
};

enum RTCRayQueryFlags { RTC_RAY_QUERY_FLAG_INCOHERENT = 0, RTC_RAY_QUERY_FLAG_COHERENT = 1 << 16 };
enum RTCFeatureFlags {
    RTC_FEATURE_FLAG_NONE = 0,
    RTC_FEATURE_FLAG_TRIANGLE = 1 << 2,
    RTC_FEATURE_FLAG_INSTANCE = 1 << 19,
    RTC_FEATURE_FLAG_ALL = 0x7fffffff
};
struct RTCIntersectArguments {
    RTCRayQueryFlags flags;
    RTCFeatureFlags feature_mask;
    void *context, *filter, *intersect;
};
struct RTCOccludedArguments {
    RTCRayQueryFlags flags;
    RTCFeatureFlags feature_mask;
    void *context, *filter, *occluded;
};
inline void rtcInitIntersectArguments(RTCIntersectArguments *a)
{
    a->flags = RTC_RAY_QUERY_FLAG_INCOHERENT;
    a->feature_mask = RTC_FEATURE_FLAG_ALL;
    a->context = a->filter = a->intersect = nullptr;
}
inline void rtcInitOccludedArguments(RTCOccludedArguments *a)
{
    a->flags = RTC_RAY_QUERY_FLAG_INCOHERENT;
    a->feature_mask = RTC_FEATURE_FLAG_ALL;
    a->context = a->filter = a->occluded = nullptr;
}

// closest hit: fills rayhit->hit and shortens rayhit->ray.tfar; a miss leaves geomID / primID invalid
void rtcIntersect1(RTCScene scene, RTCRayHit *rayhit, RTCIntersectArguments *args);
// any hit: sets ray->tfar = -inf if something lies in (tnear, tfar]
void rtcOccluded1(RTCScene scene, RTCRay *ray, RTCOccludedArguments *args);
