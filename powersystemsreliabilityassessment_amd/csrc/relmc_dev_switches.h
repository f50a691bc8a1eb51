// relmc_dev_switches.h — ONLY compiled into the -DRELMC_DEV_SWITCHES build (csrc/Makefile: ablate/librelmc_dev.so, loaded with RELMC_LIB_PATH):
// the diagnosis switches of a context and the schedule ablations read from the environment, once, for the A/B scripts under scripts/.
// The default librelmc.so does not include this file; its only environment variable is RELMC_VERBOSE.
#pragma once
#include <cstdlib>

#include "relmc_ctx.h"

inline void relmc_dev_switches_context(relmc_switches& sw)
{
    sw.no_retry = std::getenv("RELMC_NO_RETRY") != nullptr; sw.retry_dense_first = std::getenv("RELMC_RETRY_DENSE_FIRST") != nullptr;
    sw.nsq_no_stretch = std::getenv("RELMC_NSQ_NO_STRETCH") != nullptr; sw.db_no_probe = std::getenv("RELMC_DB_NO_PROBE") != nullptr;
}

inline void relmc_dev_switches_schedule(relmc_host::SymOpts& so)
{
    if (const char* q = std::getenv("RELMC_PLACE_WW")) so.place_ww = std::atol(q);
    if (const char* q = std::getenv("RELMC_PLACE_MOVES")) so.place_moves = std::atoi(q);
    so.no_quarter = std::getenv("RELMC_NO_QUARTER") != nullptr; so.no_half = std::getenv("RELMC_NO_HALF") != nullptr;
    if (const char* q = std::getenv("RELMC_SCEN_PAD4")) so.scen_pad4 = std::atoi(q);
    so.no_bwd_half = std::getenv("RELMC_NO_BWD_HALF") != nullptr; so.no_bus_map = std::getenv("RELMC_NO_BUS_MAP") != nullptr;
}
