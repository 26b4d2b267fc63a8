// The buffer surface of the C ABI: device buffers, host <-> device copies through a pinned
// staging ring owned by the context, and DLPack import / export.  With these a host binding
// needs nothing but this library to exchange arrays with numpy (host memory) and with
// frameworks that speak DLPack (tf.Tensor via tf.experimental.dlpack, torch, cupy): the
// reference's callers hand numpy arrays and tf.Tensors to this path (utils/datasets.py:130-152,
// inference.py:38-60).
#include <string.h>

#include <mutex>
#include <new>
#include <vector>

#include "ssdbox_internal.h"

namespace {

constexpr size_t kStageChunk = (size_t)4 << 20;      // bytes per staging slot
constexpr size_t kCacheLimit = (size_t)256 << 20;    // bytes of freed buffers a context keeps

// DLPack's C structures (dlpack.h, DLManagedTensor without version field), restated so that
// the library has no header dependency.
struct DLDevice {
    int32_t device_type;
    int32_t device_id;
};
struct DLDataType {
    uint8_t code;
    uint8_t bits;
    uint16_t lanes;
};
struct DLTensor {
    void *data;
    DLDevice device;
    int32_t ndim;
    DLDataType dtype;
    int64_t *shape;
    int64_t *strides;
    uint64_t byte_offset;
};
struct DLManagedTensor {
    DLTensor dl_tensor;
    void *manager_ctx;
    void (*deleter)(DLManagedTensor *self);
};
constexpr int kDLCPU = 1, kDLCUDAHost = 3, kDLROCM = 10, kDLROCMHost = 11;

// Pinned host blocks handed out by ssdbox_host_alloc: a copy to one of them needs no staging.
struct PinnedBlock {
    char *base;
    size_t bytes;
};
std::mutex g_pinned_mu;
std::vector<PinnedBlock> g_pinned;

bool is_library_pinned(const void *p, size_t bytes)
{
    const char *q = static_cast<const char *>(p);
    std::lock_guard<std::mutex> lock(g_pinned_mu);
    for (const PinnedBlock &b : g_pinned)
        if (q >= b.base && q + bytes <= b.base + b.bytes) return true;
    return false;
}

int ensure_staging(ssdbox_ctx *ctx)
{
    if (ctx->stage[0] != nullptr && ctx->stage[1] != nullptr) return SSDBOX_OK;
    for (int i = 0; i < 2; ++i) {
        hipError_t e = hipSuccess;
        if (ctx->stage[i] == nullptr) e = hipHostMalloc(&ctx->stage[i], kStageChunk, hipHostMallocDefault);
        if (e == hipSuccess && ctx->stage_ev[i] == nullptr)
            e = hipEventCreateWithFlags(&ctx->stage_ev[i], hipEventDisableTiming);
        if (e != hipSuccess) {
            // all or nothing: a later call starts over
            for (int k = 0; k < 2; ++k) {
                if (ctx->stage[k] != nullptr) (void)hipHostFree(ctx->stage[k]);
                if (ctx->stage_ev[k] != nullptr) (void)hipEventDestroy(ctx->stage_ev[k]);
                ctx->stage[k] = nullptr;
                ctx->stage_ev[k] = nullptr;
            }
            set_error("pinned staging allocation failed: %s", hipGetErrorString(e));
            return SSDBOX_ERR_NOMEM;
        }
        ctx->stage_busy[i] = false;
    }
    ctx->stage_next = 0;
    return SSDBOX_OK;
}

// Waits until the copy that last used staging slot `i` has finished.
int stage_wait(ssdbox_ctx *ctx, int i)
{
    if (ctx->stage_busy[i]) {
        SSD_CHECK_HIP(hipEventSynchronize(ctx->stage_ev[i]));
        ctx->stage_busy[i] = false;
    }
    return SSDBOX_OK;
}

int buf_alloc_impl(ssdbox_ctx *ctx, size_t bytes, void **d_out)
{
    *d_out = nullptr;
    const size_t want = bytes ? (bytes + 255) / 256 * 256 : 256;
    ctx->buf_live.reserve(ctx->buf_live.size() + 1);       // may throw here, not after the allocation
    // a freed buffer of this size, if the context kept one (reuse is in stream order: whoever
    // freed it had enqueued its last use on this context's stream)
    for (size_t i = 0; i < ctx->buf_free.size(); ++i) {
        if (ctx->buf_free[i].bytes == want) {
            *d_out = ctx->buf_free[i].ptr;
            ctx->buf_cached -= want;
            ctx->buf_free[i] = ctx->buf_free.back();
            ctx->buf_free.pop_back();
            break;
        }
    }
    if (*d_out == nullptr) {
        hipError_t e = hipMalloc(d_out, want);
        if (e != hipSuccess) {
            // give the cache back and try once more
            SSD_CHECK_HIP(hipStreamSynchronize(ctx->stream));
            for (auto &b : ctx->buf_free) (void)hipFree(b.ptr);
            ctx->buf_free.clear();
            ctx->buf_cached = 0;
            e = hipMalloc(d_out, want);
        }
        if (e != hipSuccess) {
            *d_out = nullptr;
            set_error("hipMalloc of %zu bytes failed: %s", want, hipGetErrorString(e));
            return SSDBOX_ERR_NOMEM;
        }
    }
    ctx->buf_live.push_back({*d_out, want});
    return SSDBOX_OK;
}

// Removes d from the context's live list; returns its size or 0 when it is not there.
size_t take_live(ssdbox_ctx *ctx, void *d)
{
    for (size_t i = 0; i < ctx->buf_live.size(); ++i) {
        if (ctx->buf_live[i].ptr == d) {
            const size_t bytes = ctx->buf_live[i].bytes;
            ctx->buf_live[i] = ctx->buf_live.back();
            ctx->buf_live.pop_back();
            return bytes;
        }
    }
    return 0;
}

int copy_h2d_impl(ssdbox_ctx *ctx, void *d_dst, const void *h_src, size_t bytes)
{
    int rc = ensure_staging(ctx);
    if (rc != SSDBOX_OK) return rc;
    const char *src = static_cast<const char *>(h_src);
    char *dst = static_cast<char *>(d_dst);
    for (size_t off = 0; off < bytes; off += kStageChunk) {
        const size_t n = bytes - off < kStageChunk ? bytes - off : kStageChunk;
        const int i = ctx->stage_next;
        ctx->stage_next ^= 1;
        rc = stage_wait(ctx, i);
        if (rc != SSDBOX_OK) return rc;
        memcpy(ctx->stage[i], src + off, n);
        SSD_CHECK_HIP(hipMemcpyAsync(dst + off, ctx->stage[i], n, hipMemcpyHostToDevice, ctx->stream));
        SSD_CHECK_HIP(hipEventRecord(ctx->stage_ev[i], ctx->stream));
        ctx->stage_busy[i] = true;
    }
    return SSDBOX_OK;
}

int copy_d2h_impl(ssdbox_ctx *ctx, void *h_dst, const void *d_src, size_t bytes)
{
    if (is_library_pinned(h_dst, bytes)) {         // the DMA writes the destination itself
        SSD_CHECK_HIP(hipMemcpyAsync(h_dst, d_src, bytes, hipMemcpyDeviceToHost, ctx->stream));
        SSD_CHECK_HIP(hipStreamSynchronize(ctx->stream));
        return SSDBOX_OK;
    }
    int rc = ensure_staging(ctx);
    if (rc != SSDBOX_OK) return rc;
    const char *src = static_cast<const char *>(d_src);
    char *dst = static_cast<char *>(h_dst);
    // two chunks in flight: while one is copied out of its slot the next one arrives
    size_t pend_off[2] = {0, 0}, pend_n[2] = {0, 0};
    for (size_t off = 0; off < bytes; off += kStageChunk) {
        const size_t n = bytes - off < kStageChunk ? bytes - off : kStageChunk;
        const int i = ctx->stage_next;
        ctx->stage_next ^= 1;
        rc = stage_wait(ctx, i);
        if (rc != SSDBOX_OK) return rc;
        if (pend_n[i] != 0) {
            memcpy(dst + pend_off[i], ctx->stage[i], pend_n[i]);
            pend_n[i] = 0;
        }
        SSD_CHECK_HIP(hipMemcpyAsync(ctx->stage[i], src + off, n, hipMemcpyDeviceToHost, ctx->stream));
        SSD_CHECK_HIP(hipEventRecord(ctx->stage_ev[i], ctx->stream));
        ctx->stage_busy[i] = true;
        pend_off[i] = off;
        pend_n[i] = n;
    }
    for (int k = 0; k < 2; ++k) {
        const int i = (ctx->stage_next + k) & 1;       // oldest first
        if (pend_n[i] == 0) continue;
        rc = stage_wait(ctx, i);
        if (rc != SSDBOX_OK) return rc;
        memcpy(dst + pend_off[i], ctx->stage[i], pend_n[i]);
    }
    return SSDBOX_OK;
}

// What an exported tensor carries besides the DLPack fields: its shape and the device to free on.
struct ExportHolder {
    DLManagedTensor dl;
    int64_t shape[8];
    int device;
};

void export_deleter(DLManagedTensor *self)
{
    if (self == nullptr) return;
    ExportHolder *h = static_cast<ExportHolder *>(self->manager_ctx);
    DeviceGuard guard(h->device);
    (void)hipFree(self->dl_tensor.data);
    delete h;
}

}  // namespace

void free_buffer_state(ssdbox_ctx *ctx)
{
    for (auto &b : ctx->buf_free) (void)hipFree(b.ptr);
    for (auto &b : ctx->buf_live) (void)hipFree(b.ptr);
    ctx->buf_free.clear();
    ctx->buf_live.clear();
    for (int i = 0; i < 2; ++i) {
        if (ctx->stage[i] != nullptr) (void)hipHostFree(ctx->stage[i]);
        if (ctx->stage_ev[i] != nullptr) (void)hipEventDestroy(ctx->stage_ev[i]);
        ctx->stage[i] = nullptr;
        ctx->stage_ev[i] = nullptr;
    }
}

#define SSD_NOTHROW_BEGIN try {
#define SSD_NOTHROW_END                                                     \
    }                                                                       \
    catch (const std::bad_alloc &)                                          \
    {                                                                       \
        set_error("out of host memory");                                    \
        return SSDBOX_ERR_NOMEM;                                            \
    }                                                                       \
    catch (...)                                                             \
    {                                                                       \
        set_error("unexpected C++ exception");                              \
        return SSDBOX_ERR_HIP;                                              \
    }

extern "C" int ssdbox_device_count(void)
{
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) return 0;
    return n;
}

extern "C" int ssdbox_buf_alloc(ssdbox_ctx *ctx, size_t bytes, void **d_out)
{
    SSD_REQUIRE(ctx != nullptr && d_out != nullptr, "ssdbox_buf_alloc: NULL argument");
    DeviceGuard guard(ctx->device);
    SSD_NOTHROW_BEGIN
    std::lock_guard<std::mutex> lock(ctx->buf_mu);
    return buf_alloc_impl(ctx, bytes, d_out);
    SSD_NOTHROW_END
}

extern "C" int ssdbox_buf_free(ssdbox_ctx *ctx, void *d_ptr)
{
    SSD_REQUIRE(ctx != nullptr, "ssdbox_buf_free: ctx is NULL");
    if (d_ptr == nullptr) return SSDBOX_OK;
    DeviceGuard guard(ctx->device);
    SSD_NOTHROW_BEGIN
    {
        std::lock_guard<std::mutex> lock(ctx->buf_mu);
        ctx->buf_free.reserve(ctx->buf_free.size() + 1);
        const size_t bytes = take_live(ctx, d_ptr);
        SSD_REQUIRE(bytes != 0, "ssdbox_buf_free: %p was not allocated by this context (or was exported)", d_ptr);
        if (ctx->buf_cached + bytes <= kCacheLimit) {
            ctx->buf_free.push_back({d_ptr, bytes});
            ctx->buf_cached += bytes;
            return SSDBOX_OK;
        }
    }
    SSD_CHECK_HIP(hipStreamSynchronize(ctx->stream));
    SSD_CHECK_HIP(hipFree(d_ptr));
    return SSDBOX_OK;
    SSD_NOTHROW_END
}

extern "C" int ssdbox_host_alloc(size_t bytes, void **h_out)
{
    SSD_REQUIRE(h_out != nullptr && bytes > 0, "ssdbox_host_alloc: NULL or empty");
    *h_out = nullptr;
    SSD_NOTHROW_BEGIN
    {
        std::lock_guard<std::mutex> lock(g_pinned_mu);
        g_pinned.reserve(g_pinned.size() + 1);
    }
    void *p = nullptr;
    hipError_t e = hipHostMalloc(&p, bytes, hipHostMallocPortable);
    if (e != hipSuccess) {
        set_error("hipHostMalloc of %zu bytes failed: %s", bytes, hipGetErrorString(e));
        return SSDBOX_ERR_NOMEM;
    }
    std::lock_guard<std::mutex> lock(g_pinned_mu);
    g_pinned.push_back({static_cast<char *>(p), bytes});
    *h_out = p;
    return SSDBOX_OK;
    SSD_NOTHROW_END
}

extern "C" int ssdbox_host_free(void *h_ptr)
{
    if (h_ptr == nullptr) return SSDBOX_OK;
    bool found = false;
    {
        std::lock_guard<std::mutex> lock(g_pinned_mu);
        for (size_t i = 0; i < g_pinned.size(); ++i) {
            if (g_pinned[i].base == h_ptr) {
                g_pinned[i] = g_pinned.back();
                g_pinned.pop_back();
                found = true;
                break;
            }
        }
    }
    SSD_REQUIRE(found, "ssdbox_host_free: %p did not come from ssdbox_host_alloc", h_ptr);
    SSD_CHECK_HIP(hipHostFree(h_ptr));
    return SSDBOX_OK;
}

extern "C" int ssdbox_copy_h2d(ssdbox_ctx *ctx, void *d_dst, const void *h_src, size_t bytes)
{
    SSD_REQUIRE(ctx != nullptr, "ssdbox_copy_h2d: ctx is NULL");
    if (bytes == 0) return SSDBOX_OK;
    SSD_REQUIRE(d_dst != nullptr && h_src != nullptr, "ssdbox_copy_h2d: NULL pointer");
    DeviceGuard guard(ctx->device);
    return copy_h2d_impl(ctx, d_dst, h_src, bytes);
}

extern "C" int ssdbox_copy_d2h(ssdbox_ctx *ctx, void *h_dst, const void *d_src, size_t bytes)
{
    SSD_REQUIRE(ctx != nullptr, "ssdbox_copy_d2h: ctx is NULL");
    if (bytes == 0) return SSDBOX_OK;
    SSD_REQUIRE(h_dst != nullptr && d_src != nullptr, "ssdbox_copy_d2h: NULL pointer");
    DeviceGuard guard(ctx->device);
    return copy_d2h_impl(ctx, h_dst, d_src, bytes);
}

extern "C" int ssdbox_copy_d2d(ssdbox_ctx *ctx, void *d_dst, const void *d_src, size_t bytes)
{
    SSD_REQUIRE(ctx != nullptr, "ssdbox_copy_d2d: ctx is NULL");
    if (bytes == 0) return SSDBOX_OK;
    SSD_REQUIRE(d_dst != nullptr && d_src != nullptr, "ssdbox_copy_d2d: NULL pointer");
    DeviceGuard guard(ctx->device);
    SSD_CHECK_HIP(hipMemcpyAsync(d_dst, d_src, bytes, hipMemcpyDeviceToDevice, ctx->stream));
    return SSDBOX_OK;
}

extern "C" int ssdbox_dlpack_import(ssdbox_ctx *ctx, void *dl_managed_tensor, int32_t producer_ordered, void **d_ptr,
                                    int64_t *n_elems, int32_t *dtype_code, int32_t *dtype_bits, int32_t *owned)
{
    SSD_REQUIRE(ctx != nullptr && dl_managed_tensor != nullptr && d_ptr != nullptr && n_elems != nullptr &&
                    dtype_code != nullptr && dtype_bits != nullptr && owned != nullptr,
                "ssdbox_dlpack_import: NULL argument");
    const DLTensor &t = static_cast<DLManagedTensor *>(dl_managed_tensor)->dl_tensor;
    *d_ptr = nullptr;
    *owned = 0;
    SSD_REQUIRE(t.dtype.lanes == 1 && t.dtype.bits % 8 == 0, "ssdbox_dlpack_import: vector or sub-byte dtype");
    SSD_REQUIRE(t.ndim >= 0 && t.ndim <= 8, "ssdbox_dlpack_import: %d dimensions", t.ndim);
    int64_t n = 1;
    for (int d = 0; d < t.ndim; ++d) {
        SSD_REQUIRE(t.shape[d] >= 0, "ssdbox_dlpack_import: negative extent");
        n *= t.shape[d];
    }
    if (t.strides != nullptr && n > 0) {         // must be dense row-major
        int64_t expect = 1;
        for (int d = t.ndim - 1; d >= 0; --d) {
            SSD_REQUIRE(t.shape[d] == 1 || t.strides[d] == expect,
                        "ssdbox_dlpack_import: tensor is not contiguous (make it so before the call)");
            expect *= t.shape[d];
        }
    }
    *n_elems = n;
    *dtype_code = t.dtype.code;
    *dtype_bits = t.dtype.bits;
    char *data = static_cast<char *>(t.data) + t.byte_offset;
    if (t.device.device_type == kDLROCM) {
        SSD_REQUIRE(t.device.device_id == ctx->device, "ssdbox_dlpack_import: tensor on GPU %d, context on %d",
                    t.device.device_id, ctx->device);
        // Zero copy: the caller keeps the producer alive.  The library reads the memory on the
        // context's stream, which knows nothing of the stream the producer wrote it on: unless the
        // producer has ordered its work before that stream already (the DLPack stream protocol),
        // everything pending on the device is waited for here.
        if (!producer_ordered) {
            DeviceGuard guard(ctx->device);
            SSD_CHECK_HIP(hipDeviceSynchronize());
        }
        *d_ptr = data;
        return SSDBOX_OK;
    }
    if (t.device.device_type == kDLCPU || t.device.device_type == kDLROCMHost ||
        t.device.device_type == kDLCUDAHost) {
        DeviceGuard guard(ctx->device);
        SSD_NOTHROW_BEGIN
        const size_t bytes = (size_t)n * (t.dtype.bits / 8);
        int rc;
        {
            std::lock_guard<std::mutex> lock(ctx->buf_mu);
            rc = buf_alloc_impl(ctx, bytes, d_ptr);
        }
        if (rc != SSDBOX_OK) return rc;
        *owned = 1;
        if (bytes != 0) rc = copy_h2d_impl(ctx, *d_ptr, data, bytes);
        return rc;
        SSD_NOTHROW_END
    }
    set_error("ssdbox_dlpack_import: device type %d is neither host memory nor an AMD GPU", t.device.device_type);
    return SSDBOX_ERR_UNSUPPORTED;
}

extern "C" int ssdbox_dlpack_export(ssdbox_ctx *ctx, void *d_ptr, int32_t ndim, const int64_t *shape,
                                    int32_t dtype_code, int32_t dtype_bits, void **dl_managed_tensor)
{
    SSD_REQUIRE(ctx != nullptr && d_ptr != nullptr && dl_managed_tensor != nullptr && (shape != nullptr || ndim == 0),
                "ssdbox_dlpack_export: NULL argument");
    SSD_REQUIRE(ndim >= 0 && ndim <= 8, "ssdbox_dlpack_export: %d dimensions", ndim);
    *dl_managed_tensor = nullptr;
    DeviceGuard guard(ctx->device);
    SSD_NOTHROW_BEGIN
    ExportHolder *h = new ExportHolder();
    size_t bytes;
    {
        std::lock_guard<std::mutex> lock(ctx->buf_mu);
        bytes = take_live(ctx, d_ptr);
    }
    if (bytes == 0) {
        delete h;
        set_error("ssdbox_dlpack_export: %p was not allocated by this context", d_ptr);
        return SSDBOX_ERR_INVALID;
    }
    // the consumer may use the memory on any stream: everything this context enqueued is done first
    hipError_t e = hipStreamSynchronize(ctx->stream);
    if (e != hipSuccess) {
        {
            std::lock_guard<std::mutex> lock(ctx->buf_mu);
            ctx->buf_live.push_back({d_ptr, bytes});
        }
        delete h;
        set_error("ssdbox_dlpack_export: %s", hipGetErrorString(e));
        return SSDBOX_ERR_HIP;
    }
    for (int d = 0; d < ndim; ++d) h->shape[d] = shape[d];
    h->device = ctx->device;
    h->dl.dl_tensor.data = d_ptr;
    h->dl.dl_tensor.device = {kDLROCM, ctx->device};
    h->dl.dl_tensor.ndim = ndim;
    h->dl.dl_tensor.dtype = {(uint8_t)dtype_code, (uint8_t)dtype_bits, 1};
    h->dl.dl_tensor.shape = h->shape;
    h->dl.dl_tensor.strides = nullptr;
    h->dl.dl_tensor.byte_offset = 0;
    h->dl.manager_ctx = h;
    h->dl.deleter = export_deleter;
    *dl_managed_tensor = &h->dl;
    return SSDBOX_OK;
    SSD_NOTHROW_END
}

extern "C" int ssdbox_dlpack_delete(void *dl_managed_tensor)
{
    DLManagedTensor *t = static_cast<DLManagedTensor *>(dl_managed_tensor);
    if (t != nullptr && t->deleter != nullptr) t->deleter(t);
    return SSDBOX_OK;
}
