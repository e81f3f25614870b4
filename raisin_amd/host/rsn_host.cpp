// rsn_host.cpp -- C++ host mirroring the reference's engine + CLI for the
// accelerated path (the reference host is Go: engine/engine.go, cmd/cli.go; no
// Go toolchain exists in this image, so the host above the C ABI is C++).
// It keeps what the reference keeps on the host: file I/O, the -algorithm layer
// list, .rsn naming, ratio / lossless reporting.  All codec work is librsn.
//
//   rsn -compress   <file[,file..]> [-algorithm=lzss,huffman] [-out=F | -outext=rsn] [-delete]
//   rsn -decompress <file[,file..]> [-algorithm=lzss,huffman] [-out=F | -outext=E]   [-delete=false]
//   rsn -benchmark  <file[,file..]> [-algorithm=lzss,huffman,[lzss,huffman]]
#include <algorithm>
#include <chrono>
#include <cmath>
#include <condition_variable>
#include <memory>
#include <mutex>
#include <thread>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <unistd.h>
#include <fstream>
#include <functional>
#include <map>
#include <stdexcept>
#include <string>
#include <vector>

#include "../../include/rsn.h"

using Bytes = std::vector<uint8_t>;

static Bytes read_file(const std::string &p) {
    std::ifstream f(p, std::ios::binary);
    if (!f) throw std::runtime_error("Could not open file (likely does not exist): " + p);   // cli.go:95
    return Bytes((std::istreambuf_iterator<char>(f)), std::istreambuf_iterator<char>());
}
// ioutil.WriteFile + check(err) (engine.go:194-197): a failed open, write or close is an error BEFORE
// any input is deleted (the reference panics in DecompressFile ahead of deleteFiles, cli.go:165-167)
static void write_file(const std::string &p, const Bytes &b) {
    if (p.empty()) throw std::runtime_error("refusing to write to an empty output path");
    std::ofstream f(p, std::ios::binary | std::ios::trunc);
    if (!f) throw std::runtime_error("could not open output file: " + p);
    f.write((const char *)b.data(), (std::streamsize)b.size());
    f.flush();
    if (!f) throw std::runtime_error("could not write output file: " + p);
    f.close();
    if (f.fail()) throw std::runtime_error("could not close output file: " + p);
}
// strings.TrimSuffix(f, filepath.Ext(f)) (cli.go:141-143, engine.go:177-180): the extension is the suffix from the
// final dot of the LAST path element; no dot there -> nothing is trimmed
static std::string trim_ext(const std::string &f) {
    const size_t slash = f.find_last_of('/');
    const size_t dot = f.find_last_of('.');
    if (dot == std::string::npos || (slash != std::string::npos && dot < slash)) return f;
    return f.substr(0, dot);
}

// time.Duration.String() (Go's published format: "190µs", "1.23ms" below a second, "XhYmZ.ZZZs" from a second up, fractions
// without trailing zeros, "0s" for zero) -- the timeout row (engine.go:258: ">1m0s") and the "time taken" column (engine.go:425)
static std::string go_duration_ns(long long ns) {
    if (ns == 0) return "0s";
    const bool neg = ns < 0;
    unsigned long long u = neg ? (unsigned long long)(-ns) : (unsigned long long)ns;
    auto frac = [](unsigned long long v, int prec, std::string &f) {     // fmtFrac: v / 10^prec; f = "." + the fraction's digits without trailing zeros
        unsigned long long p = 1;
        for (int k = 0; k < prec; k++) p *= 10;
        char buf[24]; snprintf(buf, sizeof buf, "%0*llu", prec, v % p);
        f = prec ? buf : "";
        while (!f.empty() && f.back() == '0') f.pop_back();
        if (!f.empty()) f = "." + f;
        return v / p;
    };
    std::string out, f;
    if (u < 1000000000ull) {
        if (u < 1000ull) out = std::to_string(u) + "ns";
        else if (u < 1000000ull) { const unsigned long long q = frac(u, 3, f); out = std::to_string(q) + f + "\xC2\xB5s"; }
        else { const unsigned long long q = frac(u, 6, f); out = std::to_string(q) + f + "ms"; }
    } else {
        const unsigned long long sec = frac(u, 9, f), mins = sec / 60;
        out = std::to_string(sec % 60) + f + "s";
        if (mins > 0) { out = std::to_string(mins % 60) + "m" + out; if (mins / 60 > 0) out = std::to_string(mins / 60) + "h" + out; }
    }
    return neg ? "-" + out : out;
}
// time.Duration.Round(m): to the nearest multiple of m, halfway values away from zero
static long long go_round_ns(long long ns, long long m) {
    const unsigned long long a = ns < 0 ? (unsigned long long)(-ns) : (unsigned long long)ns, r = a % (unsigned long long)m;
    const unsigned long long v = r + r < (unsigned long long)m ? a - r : a + (unsigned long long)m - r;
    return ns < 0 ? -(long long)v : (long long)v;
}
static std::string go_duration(long long ms) { return go_duration_ns(ms * 1000000ll); }

// check(e) -> panic in the reference; an exception here (caught per row in -benchmark, engine.go:315-328)
template <class Call>
static Bytes take(Call call) {
    uint8_t *out = nullptr; size_t n = 0;
    if (call(&out, &n) != RSN_OK) throw std::runtime_error(rsn_last_error());
    Bytes b(out, out + n);
    rsn_free(out);
    return b;
}

namespace engine {
// engine.go:101-111 Writers / :48-58 Readers, restricted to the engines on this path
using Codec = std::function<Bytes(const Bytes &)>;
static const std::map<std::string, Codec> Writers = {
    {"lzss", [](const Bytes &in) { return take([&](uint8_t **o, size_t *n) { return rsn_lzss_compress(in.data(), in.size(), RSN_LZSS_DEFAULT_WINDOW, o, n); }); }},
    {"huffman", [](const Bytes &in) { return take([&](uint8_t **o, size_t *n) { return rsn_huffman_compress(in.data(), in.size(), o, n); }); }},
};
static const std::map<std::string, Codec> Readers = {
    {"lzss", [](const Bytes &in) { return take([&](uint8_t **o, size_t *n) { return rsn_lzss_decompress(in.data(), in.size(), o, n); }); }},
    {"huffman", [](const Bytes &in) { return take([&](uint8_t **o, size_t *n) { return rsn_huffman_decompress(in.data(), in.size(), o, n); }); }},
};
static const Codec &lookup(const std::map<std::string, Codec> &m, const std::string &name) {
    auto it = m.find(name);
    if (it == m.end()) throw std::runtime_error("unknown compression engine '" + name + "' (this build carries: lzss, huffman)");
    return it->second;
}
// engine.go:443-452
static Bytes compress(Bytes content, const std::vector<std::string> &algorithms) {
    for (auto &a : algorithms) content = lookup(Writers, a)(content);
    return content;
}
// engine.go:454-479 (reverse order)
static Bytes decompress(Bytes content, const std::vector<std::string> &algorithms) {
    for (size_t i = algorithms.size(); i-- > 0;) content = lookup(Readers, algorithms[i])(content);
    return content;
}
static double entropy(const Bytes &sym, size_t total) {   // goent Entropy(p, math.Log), engine.go:410,423
    size_t cnt[256] = {0};
    for (uint8_t c : sym) cnt[c]++;
    double h = 0;
    for (size_t c : cnt) if (c) { const double p = (double)c / (double)total; h -= p * std::log(p); }
    return h;
}
struct Result { std::string engine, timeTaken; float ratio; float actualEntropy; double entropy; bool lossless, failed; };
// engine.go:357-441
static Result BenchmarkFile(const std::vector<std::string> &algorithms, const std::string &path) {
    std::string name;
    for (size_t i = 0; i < algorithms.size(); i++) name += (i ? "," : "") + algorithms[i];
    const Bytes data = read_file(path);
    const auto t0 = std::chrono::steady_clock::now();
    const Bytes c = compress(data, algorithms);
    const Bytes d = decompress(c, algorithms);
    const long long ns = std::chrono::duration_cast<std::chrono::nanoseconds>(std::chrono::steady_clock::now() - t0).count();
    const std::string tt = go_duration_ns(go_round_ns(ns, 10000));     // engine.go:425: duration.Round(10*time.Microsecond).String()
    return {name, tt, (float)c.size() / (float)data.size() * 100.f, (float)entropy(d, c.size()), entropy(data, data.size()), d == data, false};
}
}  // namespace engine

// engine/util.go: ByteCountSI (decimal units, one fractional digit)
static std::string ByteCountSI(long long b) {
    const long long unit = 1000;
    if (b < unit) return std::to_string(b) + " B";
    long long div = unit; int exp = 0;
    for (long long n = b / unit; n >= unit; n /= unit) { div *= unit; exp++; }
    char buf[32]; snprintf(buf, sizeof buf, "%.1f %cB", (double)b / (double)div, "kMGTPE"[exp]);
    return buf;
}

static std::vector<std::string> split(const std::string &s, char sep) {
    std::vector<std::string> out; std::string cur;
    for (char ch : s) { if (ch == sep) { out.push_back(cur); cur.clear(); } else if (ch != ' ') cur += ch; }
    out.push_back(cur);
    return out;
}
// cmd/cli.go:203-231
static std::vector<std::vector<std::string>> parseAlgorithms(const std::string &s) {
    std::vector<std::vector<std::string>> algs; std::vector<std::string> layer; std::string buf; bool in = false;
    for (char ch : s) {
        if (ch == ',') { if (in && !buf.empty()) layer.push_back(buf); else if (!buf.empty()) algs.push_back({buf}); buf.clear(); }
        else if (ch == '[') in = true;
        else if (ch == ']') { layer.push_back(buf); buf.clear(); in = false; algs.push_back(layer); layer.clear(); }
        else buf += ch;
    }
    if (!buf.empty()) algs.push_back({buf});
    return algs;
}

int main(int argc, char **argv) {
    std::string app = argv[0], cmd, file, algorithm, out, outext; bool has_delete = false, del = false;
    long long timeout_ms = 60000;                                   // engine.go:216; RSN_BENCH_TIMEOUT_MS shortens it for tests
    if (const char *t = getenv("RSN_BENCH_TIMEOUT_MS")) timeout_ms = atoll(t);
    for (int i = 1; i < argc; i++) {
        std::string a = argv[i];
        auto val = [&](const char *k) -> const char * { const size_t n = strlen(k); if (a.compare(0, n, k) == 0 && a.size() > n && a[n] == '=') return argv[i] + n + 1; return nullptr; };
        if (a == "-compress" || a == "-decompress" || a == "-benchmark" || a == "-help") cmd = a.substr(1);
        else if (const char *v = val("-algorithm")) algorithm = v;
        else if (const char *v = val("-out")) out = v;
        else if (const char *v = val("-outext")) outext = v;
        else if (a == "-delete") { has_delete = true; del = true; }
        else if (const char *v = val("-delete")) { has_delete = true; del = strcmp(v, "false") != 0; }
        else if (const char *v = val("-fmtduration")) { printf("%s\n", go_duration_ns(go_round_ns(atoll(v), 10000)).c_str()); return 0; }   // (tests: engine.go:425's column for a given number of nanoseconds)
        else if (a[0] != '-' && file.empty()) file = a;
    }
    if (cmd.empty()) cmd = app.size() >= 5 && app.compare(app.size() - 5, 5, "grape") == 0 ? "decompress" : "compress";   // cli.go:54-58
    if (cmd == "help" || file.empty()) {
        fprintf(stderr, "Usage: %s -compress|-decompress|-benchmark <file[,file]> [-algorithm=lzss,huffman] [-out=F] [-outext=E] [-delete]\n", argv[0]);
        return cmd == "help" ? 0 : 1;
    }
    try {
        const auto files = split(file, ',');
        if (cmd == "compress") {
            if (algorithm.empty()) algorithm = "lzss,huffman";   // reference default "lzss,arithmetic" (cli.go:99); arithmetic is not on this path
            const auto algs = split(algorithm, ',');
            auto out_name = [&](const std::string &f) { return files.size() == 1 ? (out.empty() ? f + ".rsn" : out) : f + "." + (outext.empty() ? "rsn" : outext); };   // cli.go:108-112
            auto one_data = [&](const std::string &f, const Bytes &data) {   // engine.CompressFile, engine.go:157-172, the file's bytes in hand
                const std::string o = out_name(f);
                printf("Compressing...\n");
                const Bytes c = engine::compress(data, algs);
                write_file(o, c);
                printf("Original bytes: %zu\nCompressed bytes: %zu\nCompression ratio: %.2f%%\n", data.size(), c.size(), (float)c.size() / (float)data.size() * 100.f);   // engine.go:166-169
            };
            auto one_file = [&](const std::string &f) { one_data(f, read_file(f)); };
            if (files.size() > 1 && algs.size() == 1 && algs[0] == "huffman") {
                // engine.CompressFiles loops over the files, one .rsn each (engine.go:150-154).  Independent inputs of one Huffman layer go
                // through the batch entry point -- upload, encode and download overlapped on the device (and dealt over devices with
                // RSN_BATCH_DEVICES) -- in GROUPS of at most 4 GiB, in the loop's order and with its semantics: a group's files are written
                // before the next group is read, an empty file (the reference panics, huffman.go:102) ends a group, and a group that fails
                // is done again by the loop, which stops at the failing file with everything before it on disk.
                constexpr size_t BATCH_BYTES = (size_t)4 << 30;
                size_t i = 0;
                Bytes held; bool have_held = false;                 // a file that overflowed the group before: read once, kept for this group
                while (i < files.size()) {
                    std::vector<std::string> group; std::vector<Bytes> datas; size_t size = 0;
                    bool bad = false;                               // files[i] is empty or cannot be read: the loop's turn below, in order
                    while (i < files.size()) {
                        Bytes d;
                        if (have_held) { d = std::move(held); have_held = false; }
                        else {
                            try { d = read_file(files[i]); }
                            catch (const std::exception &) { bad = true; break; }   // (the group read so far is compressed and written first, engine.go:150-154)
                        }
                        if (d.empty()) { bad = true; break; }
                        if (!group.empty() && size + d.size() > BATCH_BYTES) { held = std::move(d); have_held = true; break; }
                        size += d.size(); group.push_back(files[i]); datas.push_back(std::move(d)); i++;
                    }
                    bool done = false;
                    if (group.size() > 1) {
                        std::vector<const uint8_t *> ins; std::vector<size_t> lens;
                        for (auto &d : datas) { ins.push_back(d.data()); lens.push_back(d.size()); }
                        std::vector<uint8_t *> outs(group.size(), nullptr); std::vector<size_t> out_lens(group.size(), 0);
                        if (rsn_huffman_compress_batch(group.size(), ins.data(), lens.data(), outs.data(), out_lens.data()) == RSN_OK) {
                            for (size_t k = 0; k < group.size(); k++) {
                                printf("Compressing...\n");
                                write_file(out_name(group[k]), Bytes(outs[k], outs[k] + out_lens[k]));
                                rsn_free(outs[k]);
                                printf("Original bytes: %zu\nCompressed bytes: %zu\nCompression ratio: %.2f%%\n", datas[k].size(), out_lens[k], (float)out_lens[k] / (float)datas[k].size() * 100.f);
                            }
                            done = true;
                        }
                    }
                    if (!done) for (size_t k = 0; k < group.size(); k++) one_data(group[k], datas[k]);   // (what was read is not read again)
                    if (bad) one_file(files[i++]);                  // throws where the reference panics; everything before it is on disk
                }
            } else
            for (auto &f : files) one_file(f);
            if (has_delete && del) for (auto &f : files) remove(f.c_str());
        } else if (cmd == "decompress") {
            if (algorithm.empty()) algorithm = "lzss,huffman";
            const auto algs = split(algorithm, ',');
            for (auto &f : files) {
                std::string o = trim_ext(f);                                                        // cli.go:141-143
                if (files.size() == 1 && !out.empty()) o = out;
                if (files.size() > 1 && !outext.empty()) o = f + "." + outext;
                if (o.empty() || o == f) throw std::runtime_error("output path '" + o + "' is empty or is the input itself (" + f + "): give -out / -outext");
                printf("Decompressing...\n");
                write_file(o, engine::decompress(read_file(f), algs));
            }
            if (!has_delete || del) for (auto &f : files) remove(f.c_str());                       // -delete defaults to true (cli.go:150); only reached when every output was written
        } else {
            if (algorithm.empty()) algorithm = "lzss,huffman,[lzss,huffman]";
            // engine.go:213-309 (BenchmarkSuite) without the HTML report
            for (size_t fi = 0; fi < files.size(); fi++) {
                const std::string &f = files[fi];
                printf("Compressing file %zu/%zu - %s\n", fi + 1, files.size(), f.c_str());
                std::vector<engine::Result> done, failed;
                // engine.go:235-263: one thread per algorithm entry (goroutine), wait at most one minute, an entry that has
                // not delivered by then is a ">1m0s" DNF row and its thread is left running (detached)
                struct Slot { std::string name; bool ready = false; engine::Result r; };
                struct Shared { std::mutex mu; std::condition_variable cv; std::vector<Slot> slots; size_t pending = 0; };
                auto sh = std::make_shared<Shared>();
                const auto entries = parseAlgorithms(algorithm);
                read_file(f);                                                                       // ReadFile + check(err) before anything starts
                for (auto &algs : entries) {
                    std::string n; for (auto &a : algs) n += (n.empty() ? "" : ",") + a;
                    printf("Benchmarking %s\n", n.c_str());
                    size_t slot = sh->slots.size();
                    for (size_t k = 0; k < sh->slots.size(); k++) if (sh->slots[k].name == n) slot = k;   // resultChans is keyed by name (:240)
                    if (slot == sh->slots.size()) { sh->slots.emplace_back(); sh->slots.back().name = n; }
                    sh->pending++;
                    std::thread([sh, slot, algs, f, n]() {
                        engine::Result r;
                        try { r = engine::BenchmarkFile(algs, f); }
                        catch (const std::exception &e) { r = {n, "failed", 0, 0, 0, false, true}; }   // recover(), engine.go:315-328
                        std::lock_guard<std::mutex> lk(sh->mu);
                        if (!sh->slots[slot].ready) { sh->slots[slot].r = r; sh->slots[slot].ready = true; }
                        sh->pending--;
                        sh->cv.notify_all();
                    }).detach();
                }
                {
                    std::unique_lock<std::mutex> lk(sh->mu);
                    sh->cv.wait_for(lk, std::chrono::milliseconds(timeout_ms), [&] { return sh->pending == 0; });   // waitTimeout
                    for (auto &sl : sh->slots) {
                        if (sl.ready) (sl.r.failed ? failed : done).push_back(sl.r);
                        else failed.push_back({sl.name, ">" + go_duration(timeout_ms), 0, 0, 0, false, true});
                    }
                }
                std::stable_sort(done.begin(), done.end(), [](const engine::Result &a, const engine::Result &b) {   // engine.go:266-276
                    if (a.lossless != b.lossless) return a.lossless;
                    return a.ratio < b.ratio;
                });
                printf("%-24s | %-12s | %-17s | %-14s | %-19s | %s\n", "engine", "time taken", "compression ratio", "actual entropy", "theoretical entropy", "lossless");
                for (auto &r : done) {
                    char ratio[32]; snprintf(ratio, sizeof ratio, "%.2f%%", r.ratio);
                    printf("%-24s | %-12s | %-17s | %-14.2f | %-19.2f | %s\n", r.engine.c_str(), r.timeTaken.c_str(), ratio, r.actualEntropy, r.entropy, r.lossless ? "true" : "false");
                }
                for (auto &r : failed) printf("%-24s | %-12s | %-17s | %-14s | %-19s | %s\n", r.engine.c_str(), r.timeTaken.c_str(), "DNF", "DNF", "DNF", "false");
                printf("%-24s | %-12s | %-17s | %s\n", "File", f.c_str(), "Size", ByteCountSI((long long)read_file(f).size()).c_str());
            }
        }
    } catch (const std::exception &e) {
        fprintf(stderr, "panic: %s\n", e.what());
        fflush(stdout); fflush(stderr);
        _exit(2);                        // benchmark threads that missed the deadline may still be running (the reference's goroutines die with main too)
    }
    fflush(stdout); fflush(stderr);
    _exit(0);
}
