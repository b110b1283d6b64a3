// TEST INFRASTRUCTURE ONLY (oracle): the two cooperative-groups handles the reference uses (see cuda_runtime.h)
#pragma once
#include "cuda_runtime.h"
namespace cooperative_groups {
struct grid_group {
    unsigned long long thread_rank() const {
        hipemu::State& s = hipemu::S();
        unsigned long long block = ((unsigned long long)s.bid.z * s.grid.y + s.bid.y) * s.grid.x + s.bid.x;
        return block * s.nthreads + s.cur;
    }
};
struct thread_block {
    dim3 group_index() const { hipemu::State& s = hipemu::S(); return dim3(s.bid.x, s.bid.y, s.bid.z); }
    dim3 thread_index() const { hipemu::State& s = hipemu::S(); return dim3(s.tid.x, s.tid.y, s.tid.z); }
    unsigned thread_rank() const { return hipemu::S().cur; }
    void sync() const { hipemu::block_barrier(); }
};
static inline grid_group this_grid() { return grid_group(); }
static inline thread_block this_thread_block() { return thread_block(); }
}  // namespace cooperative_groups
