// fasta_io.hip.h — reads and reference from FASTA / FASTQ text into the arrays the C ABI takes (one concatenated base buffer,
// offsets, lengths, NUL-terminated names): the host-side entry of stage 1 (the reference hands the file names to ngmlr /
// minimap2, src/telr/TELR_alignment.py:31-51, 69-82).  Host code only; included by telr_engine.hip.
//   FASTA: record starts ('>' at the start of a line) are found by worker threads over slices of the mapped file, then every
//          record is measured and copied (line breaks dropped) in parallel.
//   FASTQ: four-line records, walked line by line ('@' may start a quality line, so records cannot be found by a scan).
//   When every sequence sits on ONE line (FASTQ; FASTA as basecallers and read simulators write it) nothing is copied: the
//   base buffer IS the mapped file and the offsets point at the sequence lines (a 30x read set: 4 GB not allocated, not
//   copied and not given back).
// Names end at the first white space, as minimap2 / ngmlr print them.
#pragma once
struct telr_fasta {
    // the base buffer is raw memory: a std::vector would zero 4 GB on one thread before the copy threads overwrite it
    char *seq = nullptr; size_t seq_bytes = 0, extent = 0; std::vector<int64_t> off; std::vector<int32_t> len;
    std::vector<char> name_buf; std::vector<const char*> names;
    void *map = nullptr; size_t map_bytes = 0;        // set: `seq` points into this mapping of the file (not owned memory)
    ~telr_fasta() { if (map) unmap_in_pieces(map, map_bytes); else free(seq); }
};
extern "C" int64_t telr_fasta_extent(const telr_fasta *f) { return f ? (int64_t)f->extent : 0; }
extern "C" void telr_fasta_free(telr_fasta *f) { delete f; }
extern "C" int32_t telr_fasta_count(const telr_fasta *f) { return f ? (int32_t)f->len.size() : 0; }
extern "C" const char *telr_fasta_seq(const telr_fasta *f) { return f ? f->seq : nullptr; }
extern "C" const int64_t *telr_fasta_off(const telr_fasta *f) { return f ? f->off.data() : nullptr; }
extern "C" const int32_t *telr_fasta_len(const telr_fasta *f) { return f ? f->len.data() : nullptr; }
extern "C" const char *const *telr_fasta_names(const telr_fasta *f) { return f ? f->names.data() : nullptr; }
extern "C" int64_t telr_fasta_bases(const telr_fasta *f) { return f ? (int64_t)f->seq_bytes : 0; }

extern "C" int telr_fasta_load(const char *path, telr_fasta **out)
{
    if (!path || !out) return TELR_E_ARG;
    // (I/O failures have their own code since round 6: TELR_E_ARG is left for a layout this parser refuses, which the Python reader may still take)
    int fd = open(path, O_RDONLY);
    if (fd < 0) return TELR_E_IO;
    struct stat sb;
    if (fstat(fd, &sb) != 0) { close(fd); return TELR_E_IO; }
    const size_t n = (size_t)sb.st_size;
    telr_fasta *F = new telr_fasta();
    if (n == 0) { close(fd); *out = F; return TELR_OK; }
    const char *p = (const char*)mmap(nullptr, n, PROT_READ, MAP_PRIVATE, fd, 0);
    close(fd);
    if (p == MAP_FAILED) { delete F; return TELR_E_IO; }
    static const bool trace = trace_on("fasta");
    auto tt0 = std::chrono::steady_clock::now();
    auto lap = [&](const char *what) { if (!trace) return; auto t1 = std::chrono::steady_clock::now(); fprintf(stderr, "[fasta %s] %-22s %8.2f ms\n", path, what, std::chrono::duration<double, std::milli>(t1 - tt0).count()); tt0 = t1; };
    struct Rec { size_t hdr, body, end; };       // header line start (after '>' / '@'), first byte after the header line, end of the sequence text
    std::vector<Rec> recs;
    const int NT = host_threads();
    auto line_end = [&](size_t i) { const char *e = (const char*)memchr(p + i, '\n', n - i); return e ? (size_t)(e - p) : n; };
    // leading blank lines / white space are not part of any record (minimap2 and ngmlr read such files)
    size_t lead = 0;
    while (lead < n && (p[lead] == '\n' || p[lead] == '\r' || p[lead] == ' ' || p[lead] == '\t')) ++lead;
    if (lead == n) { munmap((void*)p, n); *out = F; return TELR_OK; }
    if (p[lead] == '>') {
        // slices, not threads: parallel_ranges runs fewer than 64 items on the calling thread (the scan of a 4-GB file was serial)
        const int NS = std::max(64, NT * 4);
        std::vector<std::vector<size_t>> starts((size_t)NS);
        parallel_ranges(NT, NS, [&](int, int t0, int t1) {
            for (int t = t0; t < t1; ++t) {
                size_t a = n * (size_t)t / NS, b = n * (size_t)(t + 1) / NS;
                if (t == 0) starts[t].push_back(lead);
                // '>' preceded by a line break, at positions (a, b]
                for (size_t i = a; i < b; ) { const char *e = (const char*)memchr(p + i, '\n', b - i); if (!e) break; i = (size_t)(e - p) + 1; if (i < n && i > lead && p[i] == '>') starts[t].push_back(i); }
            }
        });
        std::vector<size_t> st;
        for (auto &v : starts) st.insert(st.end(), v.begin(), v.end());
        recs.resize(st.size());
        for (size_t r = 0; r < st.size(); ++r) { recs[r].hdr = st[r] + 1; recs[r].end = r + 1 < st.size() ? st[r + 1] : n; }
        parallel_ranges(NT, (int)recs.size(), [&](int, int r0, int r1) { for (int r = r0; r < r1; ++r) { size_t e = line_end(recs[r].hdr); recs[r].body = e < recs[r].end ? e + 1 : recs[r].end; } });
    } else if (p[lead] == '@') {
        // four-line records only; a multi-line FASTQ is refused here and read by the caller's own reader (fasta.load)
        for (size_t i = lead; i < n; ) {
            if (p[i] == '\n' || p[i] == '\r') { ++i; continue; }                       // blank lines between / after the records
            if (p[i] != '@') { munmap((void*)p, n); delete F; return TELR_E_ARG; }
            Rec r; r.hdr = i + 1;
            size_t e = line_end(i); r.body = e < n ? e + 1 : n;
            size_t e2 = line_end(r.body); r.end = e2;                                   // the sequence line
            size_t e3 = e2 < n ? line_end(e2 + 1) : n, e4 = e3 < n ? line_end(e3 + 1) : n;  // '+' line, quality line
            recs.push_back(r);
            i = e4 < n ? e4 + 1 : n;
        }
    } else { munmap((void*)p, n); delete F; return TELR_E_ARG; }
    lap("record starts");
    const size_t nr = recs.size();
    if (nr >= (1u << 31)) { munmap((void*)p, n); delete F; return TELR_E_RANGE; }
    F->len.resize(nr); F->off.resize(nr);
    std::vector<int64_t> nlen(nr);
    std::atomic<bool> too_long{false};
    std::atomic<int> folded{0};                // records whose sequence is not one plain line
    parallel_ranges(NT, (int)nr, [&](int, int r0, int r1) {
        int fold = 0;
        for (int r = r0; r < r1; ++r) {
            int64_t bases = 0; int lines = 0;
            for (size_t i = recs[r].body; i < recs[r].end; ) { size_t e = line_end(i); if (e > recs[r].end) e = recs[r].end; size_t l = e - i; if (l && p[e - 1] == '\r') { --l; ++fold; } bases += (int64_t)l; if (l) ++lines; i = e + 1; }
            if (lines > 1) ++fold;
            if (bases > INT32_MAX) { too_long = true; bases = 0; }
            F->len[r] = (int32_t)bases;
            size_t h = recs[r].hdr, he = recs[r].body;
            size_t k = h; while (k < he && p[k] != ' ' && p[k] != '\t' && p[k] != '\n' && p[k] != '\r') ++k;
            nlen[r] = (int64_t)(k - h);
        }
        if (fold) folded += fold;
    });
    lap("lengths + name ends");
    if (too_long.load()) { munmap((void*)p, n); delete F; return TELR_E_RANGE; }
    int64_t tot = 0, ntot = 0;
    std::vector<int64_t> noff(nr);
    for (size_t r = 0; r < nr; ++r) { F->off[r] = tot; tot += F->len[r]; noff[r] = ntot; ntot += nlen[r] + 1; }
    static const bool no_zero_copy = ab_on("fasta_copy");
    const bool in_place = folded.load() == 0 && !no_zero_copy;
    F->seq_bytes = (size_t)tot;
    if (in_place) {
        // a sequence may be empty or its line may be preceded by blank lines: the offset is where its bases start
        F->seq = (char*)p; F->extent = n; F->map = (void*)p; F->map_bytes = n;
    } else {
        F->seq = (char*)malloc((size_t)tot + 1); F->extent = (size_t)tot;
        if (!F->seq) { munmap((void*)p, n); delete F; return TELR_E_NOMEM; }
    }
    F->name_buf.resize((size_t)ntot); F->names.resize(nr);
    parallel_ranges(NT, (int)nr, [&](int, int r0, int r1) {
        for (int r = r0; r < r1; ++r) {
            if (in_place) {
                size_t i = recs[r].body;
                while (i < recs[r].end && p[i] == '\n') ++i;          // blank lines before the sequence line
                F->off[r] = (int64_t)(F->len[r] ? i : 0);
            } else {
                char *d = F->seq + F->off[r];
                for (size_t i = recs[r].body; i < recs[r].end; ) { size_t e = line_end(i); if (e > recs[r].end) e = recs[r].end; size_t l = e - i; if (l && p[e - 1] == '\r') --l; memcpy(d, p + i, l); d += l; i = e + 1; }
            }
            char *nm = F->name_buf.data() + noff[r];
            memcpy(nm, p + recs[r].hdr, (size_t)nlen[r]); nm[nlen[r]] = 0;
            F->names[r] = nm;
        }
    });
    lap("offsets + names");
    if (!in_place) munmap((void*)p, n);
    *out = F;
    return TELR_OK;
}
