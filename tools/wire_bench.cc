// wire_bench.cc -- SURVEY.md 8f rank 2: what one interval's ProcessedMetricSet costs to assemble and serialize
// for M histogram names (15 keys each with the default percentile set and the _agg_* keys).
//
//   per-key  : the reference's shape -- lh_extract, one string-keyed map insert per key (processMetrics,
//              metrics.go:483-506; addAggregates, metrics.go:590-608), then GraphiteProtocol's one %f per key
//              (graphite.go:37-48)
//   bulk     : MetricSystem::SetWireFormat(Graphite, false) -- lh_snapshot_accumulate + lh_serialize (K6):
//              keys assembled and formatted on the device, one D2H of the text
//
// Prints one JSON line.  usage: wire_bench [names=65536] [samples_per_name=16] [intervals=5]
#include "loghisto.hpp"
#include "loghisto_gpu.h"

#include <algorithm>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <string>
#include <vector>

using namespace loghisto;
using clk = std::chrono::steady_clock;

static double ms_since(clk::time_point t0) { return std::chrono::duration<double, std::milli>(clk::now() - t0).count(); }

int main(int argc, char **argv)
{
    const uint32_t M = argc > 1 ? (uint32_t)std::atoi(argv[1]) : 65536;
    const int per = argc > 2 ? std::atoi(argv[2]) : 16;
    const int intervals = argc > 3 ? std::atoi(argv[3]) : 5;
    Options opt;
    opt.max_metrics = M;
    opt.num_buffers = 2;
    MetricSystem ms(std::chrono::seconds(1), false, opt);
    std::vector<std::string> names(M);
    for (uint32_t i = 0; i < M; i++) {
        char b[32];
        std::snprintf(b, sizeof b, "svc_%05u_latency", i);
        names[i] = b;
    }
    std::vector<double> t_perkey_proc, t_perkey_ser, t_bulk, t_bulk_copy;
    size_t bytes_perkey = 0, bytes_bulk = 0, lines = 0;
    uint64_t lcg = 12345;
    for (int it = 0; it < 2 * intervals; it++) {
        const bool bulk = it & 1;
        ms.SetWireFormat(bulk ? WireFormat::Graphite : WireFormat::None, false);
        for (uint32_t i = 0; i < M; i++)
            for (int k = 0; k < per; k++) {
                lcg = lcg * 6364136223846793005ull + 1442695040888963407ull;
                ms.Histogram(names[i], 1000.0 + (double)(lcg >> 40) * (1.0 + 1e-3 * i));
            }
        auto raw = ms.collectRawMetrics();
        auto t0 = clk::now();
        auto pm = ms.processMetrics(raw);
        ms.addAggregates(raw, *pm);
        const double t_proc = ms_since(t0);
        t0 = clk::now();
        const std::string req = GraphiteProtocol(*pm);
        const double t_ser = ms_since(t0);
        raw->Release();
        if (bulk) {
            t_bulk.push_back(t_proc);
            t_bulk_copy.push_back(t_ser); // GraphiteProtocol hands out a copy of the prepared request
            bytes_bulk = req.size();
            lines = (size_t)std::count(req.begin(), req.end(), '\n');
        } else {
            t_perkey_proc.push_back(t_proc);
            t_perkey_ser.push_back(t_ser);
            bytes_perkey = req.size();
        }
    }
    auto med = [](std::vector<double> v) { std::sort(v.begin(), v.end()); return v[v.size() / 2]; };
    const double pk = med(t_perkey_proc) + med(t_perkey_ser), bk = med(t_bulk);
    std::printf("{\"names\": %u, \"samples_per_name\": %d, \"lines\": %zu, \"per_key_ms\": %.2f, "
                "\"per_key_process_ms\": %.2f, \"per_key_serialize_ms\": %.2f, \"bulk_ms\": %.2f, "
                "\"bulk_copy_out_ms\": %.2f, \"speedup\": %.1f, \"bytes_per_key\": %zu, \"bytes_bulk\": %zu, \"status\": %d}\n",
                M, per, lines, pk, med(t_perkey_proc), med(t_perkey_ser), bk, med(t_bulk_copy), pk / (bk + med(t_bulk_copy)), bytes_perkey, bytes_bulk,
                ms.last_status());
    return 0;
}
