// oval_capi_internal.h — what the other translation units of the library need
// from oval_capi.cpp: the thread-local error text behind oval_last_error() and
// the device and stream of a context.
#pragma once
#include <hip/hip_runtime.h>

#include <functional>

#include "../../include/oval_hip.h"

int oval_internal_fail(int code, const char* fmt, ...) __attribute__((format(printf, 2, 3)));
// Called inside `catch (...)` of an entry point's function-try-block: no C++ exception leaves the C ABI
// (a host in another language could only abort on one).  Records what was caught as the thread's last
// error and returns OVAL_E_NOMEM for std::bad_alloc, OVAL_E_INTERNAL for anything else.
int oval_internal_caught(void);
// The same for an entry point that hands its error text to the caller's buffer: the text goes there too.
int oval_internal_caught_text(char* err, size_t err_cap);
int oval_internal_ctx_device(const oval_ctx* ctx);
hipStream_t oval_internal_ctx_stream(const oval_ctx* ctx);
// oval_ctx_sync over the main stream and both copy streams of the context (the host-memory entry points
// leave slice copies on the latter).
int oval_internal_ctx_sync_all(oval_ctx* ctx);

// hipEventSynchronize / hipStreamSynchronize that give up: they poll, back off, and after the process-wide
// limit (oval_set_device_timeout_ms) return hipErrorNotReady with the calling thread's timed-out mark set.
// `sleepy`: nap between looks from the start (the waiter has nothing better to do than leave its core).
hipError_t oval_internal_wait_event(hipEvent_t ev, bool sleepy);
hipError_t oval_internal_wait_stream(hipStream_t stream);
bool oval_internal_wait_timed_out(bool clear);          // this thread's mark (and reset)
uint64_t oval_internal_device_timeout_ns(void);         // 0 = no limit
const char* oval_internal_wait_error(hipError_t e);     // hipGetErrorString, with the time limit spelt out

// The host-memory fill behind oval_fill_batch_host, optionally without the final
// wait: landed[k] (then body_host must be pinned and n_landed >= the number of
// slices) is recorded when slice k of the body, bytes [k * slice, (k + 1) * slice),
// has arrived in host memory.  Uses the context's own layout and staging ring,
// so the context must not start another host-memory call until the last event
// has completed.
int oval_internal_fill_host(oval_ctx* ctx, const oval_obj_desc* descs, uint64_t n_obj, void* body_host,
                            uint64_t body_bytes, hipEvent_t* landed, uint32_t n_landed, bool wait);
uint32_t oval_internal_host_slices(uint64_t span_bytes);
uint64_t oval_internal_host_slice_bytes(void);

// fn(0) .. fn(n - 1) on the library's helper threads and the calling thread;
// returns when all have run.  One such run at a time per process.
void oval_internal_parallel_for(size_t n, const std::function<void(size_t)>& fn);

// oval_pinned_alloc with host coherence asked for explicitly (whatever the process environment makes the
// default): memory a kernel writes and the host reads while that kernel is still running.
int oval_internal_pinned_alloc_coherent(oval_ctx* ctx, size_t bytes, void** hptr_out);
