// oval_capi_internal.h — what the other translation units of the library need
// from oval_capi.cpp: the thread-local error text behind oval_last_error() and
// the device and stream of a context.
#pragma once
#include <hip/hip_runtime.h>

#include "../../include/oval_hip.h"

int oval_internal_fail(int code, const char* fmt, ...) __attribute__((format(printf, 2, 3)));
int oval_internal_ctx_device(const oval_ctx* ctx);
hipStream_t oval_internal_ctx_stream(const oval_ctx* ctx);
