// host_parallel.h — the one parallel loop the host half of set_scene needs: [0, n) cut into contiguous chunks, one
// per thread, the calling thread taking the first. Results must not depend on the thread count (every use writes
// disjoint outputs that are a function of the index alone); exceptions of a worker are rethrown in the caller.
#pragma once
#include <algorithm>
#include <cstddef>
#include <exception>
#include <thread>
#include <vector>

namespace crt {

template <typename F> void parallel_for(size_t n, int threads, size_t min_per_thread, F &&body /* (begin, end) */)
{
    const size_t want = min_per_thread > 0 ? n / min_per_thread : n;
    const size_t t = std::max<size_t>(1, std::min<size_t>((size_t)std::max(1, threads), want));
    if (t <= 1) {
        body((size_t)0, n);
        return;
    }
    std::vector<std::thread> pool;
    std::vector<std::exception_ptr> errs(t);
    auto run = [&](size_t k) {
        try {
            body(n * k / t, n * (k + 1) / t);
        } catch (...) {
            errs[k] = std::current_exception();
        }
    };
    for (size_t k = 1; k < t; ++k) {
        pool.emplace_back(run, k);
    }
    run(0);
    for (std::thread &th : pool) {
        th.join();
    }
    for (const std::exception_ptr &e : errs) {
        if (e) {
            std::rethrow_exception(e);
        }
    }
}

} // namespace crt
