// Stand-in for the subset of SYCL (third-party) that the reference's embree_sycl headers and
// kernel use, so that they compile as plain C++ with g++. TEST INFRASTRUCTURE (oracle/Makefile,
// target _ref). The math functions map to <cmath> float overloads: sycl::native::* are
// reduced-precision device intrinsics in a real SYCL build; here they are libm, which is also what
// the oracle restatement uses, so the two can be compared to the last bit.
#pragma once
#include <cmath>
#include <cstdint>
#include <cstring>
#include <initializer_list>
#include <memory>

namespace sycl {

inline float fabs(float x) { return std::fabs(x); }
inline float floor(float x) { return std::floor(x); }
inline float acos(float x) { return std::acos(x); }
inline float atan2(float y, float x) { return std::atan2(y, x); }
inline float ldexp(float x, int e) { return std::ldexp(x, e); }
inline float max(float a, float b) { return a < b ? b : a; }   // SYCL: fmax-like for ordered inputs
inline float min(float a, float b) { return b < a ? b : a; }
inline uint32_t max(uint32_t a, uint32_t b) { return a < b ? b : a; }
inline uint32_t min(uint32_t a, uint32_t b) { return b < a ? b : a; }
inline int max(int a, int b) { return a < b ? b : a; }
inline int min(int a, int b) { return b < a ? b : a; }
inline int clamp(int x, int lo, int hi) { return min(max(x, lo), hi); }
inline float clamp(float x, float lo, float hi) { return min(max(x, lo), hi); }

template <typename To, typename From> inline To bit_cast(const From &from)
{
    static_assert(sizeof(To) == sizeof(From), "bit_cast size");
    To to;
    std::memcpy(&to, &from, sizeof(To));
    return to;
}

namespace native {
inline float cos(float x) { return std::cos(x); }
inline float sin(float x) { return std::sin(x); }
inline float sqrt(float x) { return std::sqrt(x); }
inline float log(float x) { return std::log(x); }
inline float powr(float x, float y) { return std::pow(x, y); }
} // namespace native

// host-side types that only appear in declarations of embree_utils.h
class device {};
class context {};
class event {};
class property_list {
public:
    property_list() = default;
    template <typename T> property_list(std::initializer_list<T>) {}
};
class queue {};
namespace usm {
enum class alloc { host, device, shared };
}
namespace ext {
namespace oneapi {
namespace property {
namespace usm {
struct device_read_only {};
} // namespace usm
} // namespace property
} // namespace oneapi
} // namespace ext

template <typename T, usm::alloc Kind> class usm_allocator : public std::allocator<T> {
public:
    template <typename U> struct rebind {
        typedef usm_allocator<U, Kind> other;
    };
    usm_allocator() = default;
    explicit usm_allocator(queue &) {}
    template <typename P> usm_allocator(queue &, std::initializer_list<P>) {}
    template <typename U> usm_allocator(const usm_allocator<U, Kind> &) {}
};

} // namespace sycl
