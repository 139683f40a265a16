// compress_thresholds_test.go -- closes the one open parity pin of loghisto_amd (DESIGN.md section 2,
// VERDICT r1 "missing" #1) on any machine that has a Go toolchain.
//
// The MI355X engine is held bit-exact to a C restatement ("the oracle") of compress
// (metrics.go:316-322) built on Go's portable math/log.go algorithm.  The reference's own
// TestCompress (metrics_test.go:151-172) only pins compress to 1 %, and the image the engine was
// built in has no Go, so whether the oracle equals REAL Go at every bucket threshold could not be
// run there.  This file is that run.
//
// Usage (inside a checkout of spacejam/loghisto, package loghisto -- compress is unexported):
//
//	cp <loghisto_amd>/integration/compress_thresholds_test.go .
//	cp <loghisto_amd>/tests/golden/thresholds_x.bin .
//	go test -run 'TestCompressThresholds|TestDecompressTable' -v
//
// thresholds_x.bin (written by tests/golden/make_thresholds.py, format documented there):
// header "LHTHRv1\0", uint32 count, uint32 reserved; then count records of
// { float64 v; int16 key; uint16 flags }, little endian.  For every extended key j = 1..70978 it
// holds the smallest non-negative v whose bucket is >= j and its two float64 neighbours (one ulp of
// v either side), with the key the oracle -- and therefore the GPU -- assigns.  flags bit 0 marks
// values beyond the int16 domain (|v| > ~2.02e142), where Go's float64->int16 conversion is
// implementation-defined; the engine reproduces amd64 (CVTTSD2SL, low 16 bits), so those records
// are checked on GOARCH=amd64 only.
//
// A PASS means: for all 212 934 values, both signs, compress(v) of the real Go build equals the key the
// engine produces -- bucket parity with metrics.go is then pinned at ulp granularity, not just defined.
package loghisto

import (
	"encoding/binary"
	"math"
	"os"
	"runtime"
	"testing"
)

func thresholdsPath() string {
	if p := os.Getenv("LOGHISTO_THRESHOLDS"); p != "" {
		return p
	}
	return "thresholds_x.bin"
}

func TestCompressThresholds(t *testing.T) {
	raw, err := os.ReadFile(thresholdsPath())
	if err != nil {
		t.Skipf("fixture not found (%v): copy tests/golden/thresholds_x.bin next to this file "+
			"or set LOGHISTO_THRESHOLDS", err)
	}
	if len(raw) < 16 || string(raw[:8]) != "LHTHRv1\x00" {
		t.Fatalf("bad fixture header")
	}
	n := int(binary.LittleEndian.Uint32(raw[8:12]))
	if len(raw) != 16+12*n {
		t.Fatalf("fixture holds %d bytes, header says %d records", len(raw), n)
	}
	amd64 := runtime.GOARCH == "amd64"
	checked, skipped, bad := 0, 0, 0
	for i := 0; i < n; i++ {
		rec := raw[16+12*i : 16+12*i+12]
		v := math.Float64frombits(binary.LittleEndian.Uint64(rec[0:8]))
		want := int16(binary.LittleEndian.Uint16(rec[8:10]))
		flags := binary.LittleEndian.Uint16(rec[10:12])
		if flags&1 != 0 && !amd64 {
			skipped++
			continue
		}
		checked++
		if got := compress(v); got != want {
			bad++
			if bad <= 20 {
				t.Errorf("compress(%v [bits %#x]) = %d, engine/oracle say %d (record %d, flags %d)",
					v, math.Float64bits(v), got, want, i, flags)
			}
		}
		// metrics.go:318-320: negative values return -1 * i in int16 arithmetic
		neg := -1 * want
		if got := compress(-v); got != neg && v != 0 {
			bad++
			if bad <= 20 {
				t.Errorf("compress(%v) = %d, engine/oracle say %d (record %d)", -v, got, neg, i)
			}
		}
	}
	t.Logf("%d records checked (both signs), %d out-of-int16-domain records skipped on %s, %d mismatches",
		checked, skipped, runtime.GOARCH, bad)
	if bad != 0 {
		t.Fatalf("%d of %d threshold neighbours are bucketed differently by this Go build (%s) than by "+
			"loghisto_amd's oracle: report the Go version; the engine's threshold table must then be "+
			"regenerated from this build's math.Log", bad, 2*checked, runtime.Version())
	}
}

// TestDecompressTable compares a checksum of decompress over all 65 536 keys with the oracle's
// (`python -c "import oracle, zlib; print(zlib.crc32(oracle.decompress_table().tobytes()))"`;
// table order: bin = uint16(key) ^ 0x8000, i.e. ascending key).  decompress is already pinned bit-for-bit
// by the 15 doc goldens (readme.md:35-43, print_benchmark.go:34-39); this extends the pin to every key.
func TestDecompressTable(t *testing.T) {
	crc := uint32(0xffffffff)
	var table [256]uint32
	for i := range table {
		c := uint32(i)
		for k := 0; k < 8; k++ {
			if c&1 != 0 {
				c = 0xedb88320 ^ (c >> 1)
			} else {
				c >>= 1
			}
		}
		table[i] = c
	}
	var buf [8]byte
	for k := -32768; k <= 32767; k++ {
		binary.LittleEndian.PutUint64(buf[:], math.Float64bits(decompress(int16(k))))
		for _, b := range buf {
			crc = table[byte(crc)^b] ^ (crc >> 8)
		}
	}
	t.Logf("crc32(decompress table, ascending key, little-endian float64) = %d", crc^0xffffffff)
	const oracleCRC = uint32(1523530441) // zlib.crc32(oracle.decompress_table().tobytes()), tests/test_oracle.py
	if crc^0xffffffff != oracleCRC {
		t.Fatalf("decompress table crc %d != the oracle's %d: some decompress(k) differs from this Go build (%s)",
			crc^0xffffffff, oracleCRC, runtime.Version())
	}
}
