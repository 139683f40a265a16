// lhgpu.go -- cgo binding of liblhgpu.so for package loghisto (spacejam/loghisto).
//
// NOT COMPILED in the build image (no Go toolchain there); assembled from the snippets of
// INTEGRATION.md so that a maintainer can drop it next to metrics.go.  The tested twin of this file
// is the C++ host layer (include/loghisto.hpp, loghisto_amd/csrc/host/metric_system.cc).
//
// What changes in metrics.go itself:
//   * MetricSystem gains a field   gpu *gpuEngine   (created in NewMetricSystem, destroyed in Stop)
//   * RawMetricSet gains a field   snap *C.lh_snapshot
//   * the bodies of Histogram, the histogram part of collectRawMetrics and processHistograms are
//     replaced by the functions below (fragments marked "in collectRawMetrics" are statements to paste).

package loghisto

/*
#cgo CFLAGS: -I${SRCDIR}/include
#cgo LDFLAGS: -L${SRCDIR}/lib -llhgpu -Wl,-rpath,${SRCDIR}/lib
#include <stdlib.h>
#include "loghisto_gpu.h"
*/
import "C"

import (
	"runtime"
	"sync"
	"unsafe"

	"github.com/golang/glog"
)

const stageCap = 4096 // samples per crossing: a cgo call costs more than the old fast path

// stage is a per-P (per logical processor) staging buffer; sync.Pool keeps them P-local.  It lives in GO memory: a
// stage may sit idle in the pool for any length of time, be dropped by the GC or multiply when a P's slot is empty,
// and none of that may hold engine resources.  (Round 3 made a stage a reservation of pinned staging memory held
// across Put: stages that the pool dropped kept their lanes reserved, and once every lane was taken the next refill
// blocked inside lh_reserve_pairs under histogramMu.RLock -- the flip could never take the write lock again.  ADVICE
// r3.)  The pinned memory is touched in ship() only: reserve, copy, commit, all inside one call, exactly what the
// tested C++ twin does (MetricSystem::ship, loghisto_amd/csrc/host/metric_system.cc).
type stage struct {
	ids  [stageCap]uint32
	vals [stageCap]float64
	n    int
}

type gpuEngine struct {
	e      *C.lh_engine
	idsMu  sync.RWMutex
	ids    map[string]uint32 // name -> dense id (cache of lh_intern)
	names  []string
	pool   sync.Pool  // *stage
	all    []*stage   // every stage ever handed out, for the flush at the flip
	allMu  sync.Mutex
	narrow bool       // <= 65 536 names: ids cross PCIe as uint16 (lh_reserve_pairs16: 10 B per sample, not 12)
	decompress [65536]float64 // decompress(key) by dense bin uint16(key) ^ 0x8000, as the device generated it (lh_codec_tables)
}

func newGPUEngine(maxMetrics int) *gpuEngine {
	var cfg C.lh_config
	C.lh_default_config(&cfg)
	cfg.max_metrics = C.uint32_t(maxMetrics)
	cfg.num_lanes = C.uint32_t(runtime.GOMAXPROCS(0)) // concurrent ship() calls: at most one per P
	// cfg.cell_bits stays 0 (ABI 7): uint64 cells up to 8 192 names, uint32 above -- exact either way (the engine moves an
	// epoch buffer to uint64 cells before its interval could hold 2^32 samples); 64 pins the reference's cell width
	g := &gpuEngine{ids: make(map[string]uint32), narrow: maxMetrics <= 65536}
	if rc := C.lh_create(&cfg, &g.e); rc != C.LH_OK {
		glog.Errorf("lh_create: %s (%s)", C.GoString(C.lh_strerror(rc)), C.GoString(C.lh_last_error()))
		return nil // caller falls back to refusing Histogram; there is no CPU path in the library
	}
	// the decompress table, once: what a compact extract's keys are turned into values with (processHistogramsGPU)
	if rc := C.lh_codec_tables(g.e, nil, (*C.double)(unsafe.Pointer(&g.decompress[0]))); rc != C.LH_OK {
		glog.Errorf("lh_codec_tables: %s", C.GoString(C.lh_strerror(rc)))
	}
	g.pool.New = func() interface{} {
		s := new(stage)
		g.allMu.Lock(); g.all = append(g.all, s); g.allMu.Unlock()
		return s
	}
	return g
}

func (g *gpuEngine) id(name string) uint32 {
	g.idsMu.RLock()
	id, ok := g.ids[name]
	g.idsMu.RUnlock()
	if ok {
		return id
	}
	var cid C.uint32_t
	cs := C.CString(name)
	rc := C.lh_intern(g.e, cs, C.size_t(len(name)), &cid)
	C.free(unsafe.Pointer(cs))
	if rc != C.LH_OK {
		glog.Errorf("lh_intern(%q): %s", name, C.GoString(C.lh_strerror(rc)))
		return ^uint32(0)
	}
	g.idsMu.Lock()
	g.ids[name] = uint32(cid)
	for len(g.names) <= int(cid) { g.names = append(g.names, "") }
	g.names[cid] = name
	g.idsMu.Unlock()
	return uint32(cid)
}

// ship moves what the stage holds into the engine's pinned staging memory: reserve -> copy -> commit, never holding
// the reservation across anything that can block or across a return.  The copy is the sample's only store into pinned
// memory; the ingest kernel reads it in place over PCIe.  With at most 65 536 names the ids are narrowed to uint16 on
// the way (SURVEY.md 8d: 10 B per sample).  No Go pointer reaches C: C hands out its own memory and Go writes into it.
func (g *gpuEngine) ship(s *stage) {
	for done := 0; done < s.n; {
		var pv *C.double
		var granted C.size_t
		var tok C.uint32_t
		want := C.size_t(s.n - done)
		if g.narrow {
			var pi *C.uint16_t
			if rc := C.lh_reserve_pairs16(g.e, want, &pi, &pv, &granted, &tok); rc != C.LH_OK {
				glog.Errorf("lh_reserve_pairs16: %s", C.GoString(C.lh_strerror(rc)))
				break // the engine is unusable (device lost): the interval's remaining samples are dropped WITH a log line
			}
			ids, vals := unsafe.Slice(pi, int(granted)), unsafe.Slice(pv, int(granted))
			for i := range ids {
				ids[i] = C.uint16_t(s.ids[done+i])
				vals[i] = C.double(s.vals[done+i])
			}
		} else {
			var pi *C.uint32_t
			if rc := C.lh_reserve_pairs(g.e, want, &pi, &pv, &granted, &tok); rc != C.LH_OK {
				glog.Errorf("lh_reserve_pairs: %s", C.GoString(C.lh_strerror(rc)))
				break
			}
			ids, vals := unsafe.Slice(pi, int(granted)), unsafe.Slice(pv, int(granted))
			for i := range ids {
				ids[i] = C.uint32_t(s.ids[done+i])
				vals[i] = C.double(s.vals[done+i])
			}
		}
		if rc := C.lh_commit_pairs(g.e, tok, granted); rc != C.LH_OK { // (== lh_commit_pairs16: the token knows the width)
			glog.Errorf("lh_commit_pairs: %s", C.GoString(C.lh_strerror(rc)))
		}
		done += int(granted)
	}
	s.n = 0
}

func (ms *MetricSystem) Histogram(name string, value float64) {
	g := ms.gpu
	id := g.id(name)
	if id == ^uint32(0) { return }
	ms.histogramMu.RLock()          // same lock, same role: readers = submitters, writer = the flip
	s := g.pool.Get().(*stage)
	s.ids[s.n] = id                 // Go memory; compress() now happens on the GPU
	s.vals[s.n] = value
	s.n++
	if s.n == stageCap { g.ship(s) } // one cgo crossing per stageCap samples
	g.pool.Put(s)
	ms.histogramMu.RUnlock()
}

	ms.histogramMu.Lock()                 // excludes submitters exactly as the map swap did
	for _, s := range ms.gpu.all { ms.gpu.ship(s) }
	var snap *C.lh_snapshot
	rc := C.lh_flip(ms.gpu.e, &snap)      // LH_EBUSY: two intervals still being processed;
	ms.histogramMu.Unlock()               // the epoch keeps accumulating, nothing is lost
	if rc != C.LH_OK {
		glog.Errorf("lh_flip: %s", C.GoString(C.lh_strerror(rc)))
	}
	// RawMetricSet gains an unexported `snap *C.lh_snapshot`; Histograms is filled with
	// lh_buckets only when len(ms.rawSubscribers) > 0.

func (ms *MetricSystem) processHistogramsGPU(raw *RawMetricSet, out map[string]float64) {
	labels := make([]string, 0, len(ms.percentiles))
	ps := make([]C.double, 0, len(ms.percentiles))
	for l, p := range ms.percentiles { labels = append(labels, l); ps = append(ps, C.double(p)) }
	n := len(ms.gpu.names)
	if raw.snap == nil || n == 0 { return }
	// The COMPACT results (ABI 6, lh_extract_rows_compact): count, sum, the selected keys and a word of valid bits per
	// name, in place in the engine's pinned block -- 42 B per name instead of 139 B.  Everything else processHistograms
	// emits is a function of these: _avg = sum / float64(count) (metrics.go:356), the lifetime add uint64(sum)
	// (metrics.go:374), a percentile's value = decompress(key) (metrics.go:326-332) = the device-generated table
	// D[] that newGPUEngine read once with lh_codec_tables.
	var c C.lh_extract_compact
	rc := C.lh_extract_rows_compact(raw.snap, 0, C.size_t(n), &ps[0], C.size_t(len(ps)), &c)
	if rc != C.LH_OK {
		C.lh_release(raw.snap)
		glog.Errorf("lh_extract_rows_compact: %s", C.GoString(C.lh_strerror(rc)))
		return
	}
	np := len(ps)
	count := unsafe.Slice((*uint64)(unsafe.Pointer(c.count)), n) // views of C memory: valid until the next
	sum := unsafe.Slice((*float64)(unsafe.Pointer(c.sum)), n)     // lh_extract* / lh_release on this engine
	keys := unsafe.Slice((*int16)(unsafe.Pointer(c.pkeys)), n*np)
	valid := unsafe.Slice((*uint32)(unsafe.Pointer(c.pvalid_bits)), n)
	D := &ms.gpu.decompress // [65536]float64, indexed by the dense bin uint16(key) ^ 0x8000
	for id, name := range ms.gpu.names {
		if count[id] == 0 { continue }            // name absent from this interval
		out[name+"_count"] = float64(count[id])
		out[name+"_sum"] = sum[id]
		out[name+"_avg"] = sum[id] / float64(count[id])
		ms.addAggregates(name, uint64(sum[id]), count[id]) // metrics.go:359-376
		for i, l := range labels {
			if valid[id]>>uint(i)&1 != 0 {
				out[fmt.Sprintf(l, name)] = D[uint16(keys[id*np+i])^0x8000]
			} else {
				glog.Errorf("unable to calculate percentile: %s", "Invalid percentile.  Should be between 0 and 1.")
			}
		}
	}
	C.lh_release(raw.snap)
}

func (raw *RawMetricSet) fillHistograms(g *gpuEngine) {
	n := len(g.names)
	offsets := make([]C.uint64_t, n+1)
	var total C.size_t
	C.lh_buckets_all(raw.snap, 0, C.size_t(n), &offsets[0], nil, nil, 0, &total)   // sizes only
	if total == 0 { return }
	keys := make([]C.int16_t, total)
	counts := make([]C.uint64_t, total)
	C.lh_buckets_all(raw.snap, 0, C.size_t(n), &offsets[0], &keys[0], &counts[0], total, &total)
	for id, name := range g.names {
		if offsets[id] == offsets[id+1] { continue }            // no map entry this interval
		m := make(map[int16]*uint64, offsets[id+1]-offsets[id])
		for i := offsets[id]; i < offsets[id+1]; i++ { c := uint64(counts[i]); m[int16(keys[i])] = &c }
		raw.Histograms[name] = m
	}
}

// Bulk wire text (K6): the interval's histogram keys as a finished Graphite request, assembled and
// formatted on the device; replaces the per-key loops of metrics.go:483-506, 590-608 and graphite.go:37-48.
func (ms *MetricSystem) graphiteRequest(raw *RawMetricSet, host string) []byte {
	C.lh_snapshot_accumulate(raw.snap) // histogramCountStore += ..., metrics.go:359-376
	labels, ps := ms.percentileLabelsAndValues()
	clabels := make([]*C.char, len(labels))
	for i, l := range labels {
		clabels[i] = C.CString(l)
		defer C.free(unsafe.Pointer(clabels[i]))
	}
	ts := strconv.FormatInt(raw.Time.Unix(), 10)
	f := C.lh_line_format{
		prefix: C.CString("cockroach." + host + "."), sep: C.CString(" "), suffix: C.CString(" " + ts + "\n"),
		flags: C.LH_FMT_UNDERSCORE_TO_DOT,
	}
	defer C.free(unsafe.Pointer(f.prefix))
	defer C.free(unsafe.Pointer(f.sep))
	defer C.free(unsafe.Pointer(f.suffix))
	buf := make([]byte, ms.lastRequestBytes+ms.lastRequestBytes/8+4096)
	var n C.size_t
	for {
		C.lh_serialize(raw.snap, 0, C.size_t(len(ms.gpu.names)), (*C.double)(&ps[0]), &clabels[0],
			C.size_t(len(ps)), &f, C.LH_SER_AGGREGATES, (*C.char)(unsafe.Pointer(&buf[0])), C.size_t(len(buf)), &n)
		if int(n) <= len(buf) {
			break
		}
		buf = make([]byte, int(n)+int(n)/8) // *len reports the need; nothing was written
	}
	ms.lastRequestBytes = int(n)
	return append(buf[:n], GraphiteProtocol(ms.hostSideKeys(raw))...) // counters, rates, gauges
}

	// once: comm is an ncclComm_t from RCCL's own cgo binding; tell the library which RCCL that is
	C.lh_set_rccl_library(C.CString("librccl.so"))
	...
	// in collectRawMetrics, right after lh_flip, before anything reads the snapshot
	var first, last C.uint32_t
	rc = C.lh_snapshot_merge(snap, unsafe.Pointer(comm), C.int(nranks), C.int(rank),
		C.LH_MERGE_REDUCE_SCATTER, C.uint32_t(len(ms.gpu.names)), &first, &last)
	// this rank now holds the merged cells of names [first, last): extract only those
	C.lh_extract_rows(snap, first, C.size_t(last-first), &ps[0], C.size_t(len(ps)), &stats[0], &pvals[0], nil, &pvalid[0])

// ---- round 2 (ABI version 2) -------------------------------------------------------------------------------------

// Counter on the device (optional; replaces metrics.go:251-269 and the counter half of collectRawMetrics,
// metrics.go:425-458).  The per-P stage gains cids / camts / cn beside ids / vals / n.
func (ms *MetricSystem) counterGPU(name string, amount uint64) {
	g := ms.gpu
	id := g.counterID(name) // cache over lh_intern_counter: counters have their own dense id space
	ms.counterMu.RLock()
	s := g.pool.Get().(*stage)
	s.cids[s.cn] = C.uint32_t(id)
	s.camts[s.cn] = C.uint64_t(amount)
	s.cn++
	if s.cn == stageCap {
		C.lh_submit_counts(g.e, &s.cids[0], &s.camts[0], C.size_t(s.cn)) // copies before returning
		s.cn = 0
	}
	g.pool.Put(s)
	ms.counterMu.RUnlock()
}

func (raw *RawMetricSet) fillCounters(g *gpuEngine) {
	n := len(g.counterNames)
	if n == 0 {
		return
	}
	rate := make([]C.uint64_t, n)
	total := make([]C.uint64_t, n)
	present := make([]C.uint8_t, n)
	known := make([]C.uint8_t, n)
	C.lh_counters_collect(raw.snap, 0, C.size_t(n), &rate[0], &present[0], &total[0], &known[0])
	for id, name := range g.counterNames {
		if present[id] != 0 {
			raw.Rates[name] = uint64(rate[id]) // metrics.go:430-433
		}
		if known[id] != 0 {
			raw.Counters[name] = uint64(total[id]) // metrics.go:435-458
		}
	}
}

// Results in place for large name spaces: the library's own pinned arrays.  Every call that returns results through
// the engine's pinned block invalidates them (lh_extract*, lh_buckets*, lh_serialize*, lh_counters_collect,
// lh_lifetime, lh_format_f, lh_snapshot_merge) and so does lh_release: the caller consumes the slices (or copies
// what it keeps) BEFORE it asks for counters or lifetime totals of the same interval.
func (ms *MetricSystem) extractView(raw *RawMetricSet, ps []C.double) (stats []C.lh_stats, pvals []C.double) {
	n := len(ms.gpu.names)
	var v C.lh_extract_view
	if rc := C.lh_extract_rows_view(raw.snap, 0, C.size_t(n), &ps[0], C.size_t(len(ps)), &v); rc != C.LH_OK {
		glog.Errorf("lh_extract_rows_view: %s", C.GoString(C.lh_strerror(rc)))
		return nil, nil
	}
	return unsafe.Slice(v.stats, n), unsafe.Slice(v.pvals, n*len(ps))
}

// Self-metrics of the engine as gauges (RegisterGaugeFunc, metrics.go:299).
func (ms *MetricSystem) registerEngineGauges() {
	get := func(f func(c *C.lh_counters) float64) func() float64 {
		return func() float64 { var c C.lh_counters; C.lh_get_counters(ms.gpu.e, &c); return f(&c) }
	}
	ms.RegisterGaugeFunc("lhgpu.samples_partitioned", get(func(c *C.lh_counters) float64 { return float64(c.samples_partitioned) }))
	ms.RegisterGaugeFunc("lhgpu.scratch_bytes", get(func(c *C.lh_counters) float64 { return float64(c.scratch_bytes) }))
	ms.RegisterGaugeFunc("lhgpu.region_overflows", get(func(c *C.lh_counters) float64 { return float64(c.region_overflows) }))
	ms.RegisterGaugeFunc("lhgpu.flips_busy", get(func(c *C.lh_counters) float64 { return float64(c.flips_busy) }))
}

