// TEST INFRASTRUCTURE ONLY -- never linked into or called from the product path.
//
// Thin command-line driver around the *reference* ExactOverlapper
// (/root/reference/src/overlapper.{h,cpp}, compiled where it lies by
// oracle/Makefile into oracle/_ref/).  This file is new code: it only calls the
// reference's public C++ interface (overlapper.h:19-25) the same way the
// pybind11 glue (src/phasm.cpp:12-15) and the CLI (phasm/cli/assembler.py:29-50)
// do.  It exists to (a) generate golden vectors for tests/golden/ and (b) be the
// "reference" CPU baseline timed by bench.py.
//
// Input  (stdin or file):  one read per line:  <id> <TAB or space> <sequence>
//                          an empty sequence is written as "-"? no: "<id>\t" with nothing after
// Output (stdout):         one row per line:   a_id \t b_id \t astart \t aend \t bstart \t bend
// stderr:                  "ref_rows=<n> ref_seconds=<wall seconds of overlaps()>"
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <fstream>
#include <iostream>
#include <string>

#include "overlapper.h"

int main(int argc, char** argv) {
    if (argc < 2) {
        std::fprintf(stderr, "usage: %s <min_length> [reads.tsv] [--quiet]\n", argv[0]);
        return 2;
    }
    unsigned int min_length = static_cast<unsigned int>(std::strtoul(argv[1], nullptr, 10));
    bool quiet = false;
    const char* path = nullptr;
    for (int i = 2; i < argc; ++i) {
        if (std::string(argv[i]) == "--quiet") quiet = true; else path = argv[i];
    }
    std::ifstream fin;
    if (path) {
        fin.open(path);
        if (!fin) { std::fprintf(stderr, "cannot open %s\n", path); return 2; }
    }
    std::istream& in = path ? static_cast<std::istream&>(fin) : std::cin;

    ExactOverlapper ov;
    std::string line;
    size_t n = 0;
    while (std::getline(in, line)) {
        if (line.empty()) continue;
        size_t sep = line.find_first_of("\t ");
        std::string id = line.substr(0, sep);
        std::string seq = sep == std::string::npos ? std::string() : line.substr(sep + 1);
        ov.addSequence(id, seq);
        ++n;
    }
    if (n == 0) {  // the reference's iterator over an empty index is undefined behaviour
        std::fprintf(stderr, "ref_rows=0 ref_seconds=0\n");
        return 0;
    }
    auto t0 = std::chrono::steady_clock::now();
    std::vector<OverlapT> rows = ov.overlaps(min_length);
    auto t1 = std::chrono::steady_clock::now();
    double secs = std::chrono::duration<double>(t1 - t0).count();
    if (!quiet) {
        for (const auto& r : rows) {
            std::printf("%s\t%s\t%d\t%d\t%d\t%d\n", std::get<0>(r).c_str(), std::get<1>(r).c_str(),
                        std::get<2>(r), std::get<3>(r), std::get<4>(r), std::get<5>(r));
        }
    }
    std::fprintf(stderr, "ref_rows=%zu ref_seconds=%.6f\n", rows.size(), secs);
    return 0;
}
