// Launch timing with hipEvents on the launch stream (measurement aid for bench.py; see sdn_common.h).
#include <mutex>
#include <vector>

#include "sdn_common.h"

namespace sdn {

struct TimedRec {
    hipEvent_t e0, e1;
    double work;
};
static std::mutex g_mu;
static bool g_on = false;
static std::vector<TimedRec> g_recs[TIME_SLOTS];
static thread_local double t_declared = 0.0;

void timing_declare_work(double work) { t_declared = work > 0.0 ? work : 0.0; }

TimedLaunch::TimedLaunch(int slot, hipStream_t st, double work) : slot_(slot), st_(st), work_(work)
{
    if (t_declared > 0.0) {   // the caller's figure (true channel counts) replaces the launcher's padded one
        work_ = t_declared;
        t_declared = 0.0;
    }
    bool on;
    {
        std::lock_guard<std::mutex> lk(g_mu);
        on = g_on;
    }
    if (!on) return;
    if (hipEventCreate(&e0_) != hipSuccess || hipEventCreate(&e1_) != hipSuccess) {
        e0_ = e1_ = nullptr;
        return;
    }
    (void)hipEventRecord(e0_, st_);
}

TimedLaunch::~TimedLaunch()
{
    if (!e0_) return;
    (void)hipEventRecord(e1_, st_);
    std::lock_guard<std::mutex> lk(g_mu);
    g_recs[slot_].push_back(TimedRec{e0_, e1_, work_});
}

static int read_slot(int slot, double* ms_total, long* launches, double* work)
{
    std::vector<TimedRec> ev;
    {
        std::lock_guard<std::mutex> lk(g_mu);
        ev.swap(g_recs[slot]);
    }
    double total = 0, w = 0;
    for (auto& r : ev) {
        float ms = 0;
        if (hipEventSynchronize(r.e1) != hipSuccess || hipEventElapsedTime(&ms, r.e0, r.e1) != hipSuccess)
            return fail(SDN_ELAUNCH, "sdn_timing_read: event query failed");
        total += ms;
        w += r.work;
        (void)hipEventDestroy(r.e0);
        (void)hipEventDestroy(r.e1);
    }
    *ms_total = total;
    *launches = (long)ev.size();
    if (work) *work = w;
    return SDN_OK;
}

}  // namespace sdn

using namespace sdn;

SDN_API int sdn_timing_enable(int enable)
{
    std::lock_guard<std::mutex> lk(g_mu);
    g_on = enable != 0;
    return SDN_OK;
}

SDN_API int sdn_timing_declare_work(double work)
{
    timing_declare_work(work);
    return SDN_OK;
}

SDN_API int sdn_timing_read(double* ms_total, long* launches)
{
    if (!ms_total || !launches) return fail(SDN_EINVAL, "sdn_timing_read: null result slot");
    return read_slot(TIME_RASTER_TILES, ms_total, launches, nullptr);
}

SDN_API int sdn_timing_read_slot(int slot, double* ms_total, long* launches, double* work)
{
    if (!ms_total || !launches) return fail(SDN_EINVAL, "sdn_timing_read_slot: null result slot");
    if (slot < 0 || slot >= TIME_SLOTS) return fail(SDN_EINVAL, "sdn_timing_read_slot: slot %d", slot);
    return read_slot(slot, ms_total, launches, work);
}
