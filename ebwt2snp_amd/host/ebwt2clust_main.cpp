// ebwt2clust -- drop-in for the reference executable (ebwt2clust.cpp:141-198): same options, same
// defaults (0 = default), same input discovery, same <input>.clusters output and stdout lines.
// The scan itself (cluster_lm, ebwt2clust.cpp:68-139) runs on the GPU through the C ABI.
#include <fcntl.h>
#include <unistd.h>

#include <fstream>
#include <vector>

#include "egsa_files.hpp"

static const int min_def = 2, K_def = 16, lcp_def = 1, da_def = 4, pos_def = 1;

static void help() {
    std::cout << "ebwt2clust [options]\nOptions:\n"
                 "-h         Print this help\n"
                 "-i <arg>   Input fasta file (REQUIRED)\n"
                 "-k <arg>   Minimum LCP required in clusters (default: " << K_def << ")\n"
                 "-m <arg>   Discard clusters smaller than this value (default: " << min_def << ")\n"
                 "-x <arg>   Byte size of LCP integers in input EGSA/BCR file (default: " << lcp_def << ").\n"
                 "-y <arg>   Byte size of DA integers (read number) in input EGSA/BCR file (default: " << da_def << ").\n"
                 "-z <arg>   Byte size of pos integers (position in read) in input EGSA/BCR file (default: " << pos_def << ").\n\n"
                 "\nThe Enhanced Generalized Suffix Array of the input must exist next to it, either as <input>.gesa\n"
                 "(github.com/felipelouza/egsa) or as the BCR triple <input>.out, <input>.out.lcp, <input>.out.pairSA\n"
                 "(github.com/giovannarosone/BCR_LCP_GSA). Output is stored in <input>.clusters.\n"
                 "This build runs the scan on an AMD MI355X (gfx950); extra option: -G <gpu id> (default 0).\n";
    exit(0);
}

int main(int argc, char** argv) {
    if (argc < 2) help();
    int k = 0, min_len = 0, lcp = 0, da = 0, pos = 0, gpu = 0;
    std::string input;
    int opt;
    while ((opt = getopt(argc, argv, "hk:i:m:x:y:z:G:")) != -1) {
        switch (opt) {
            case 'h': help(); break;
            case 'k': k = atoi(optarg); break;
            case 'm': min_len = atoi(optarg); break;
            case 'i': input = std::string(optarg); break;
            case 'x': lcp = atoi(optarg); break;
            case 'y': da = atoi(optarg); break;
            case 'z': pos = atoi(optarg); break;
            case 'G': gpu = atoi(optarg); break;
            default: help(); return -1;
        }
    }
    lcp = lcp == 0 ? lcp_def : lcp;
    da = da == 0 ? da_def : da;
    pos = pos == 0 ? pos_def : pos;
    k = k == 0 ? K_def : k;
    min_len = min_len == 0 ? min_def : min_len;
    if (input.compare("") == 0) help();

    PhaseClock pc;
    CliReport report;
    EgsaFiles F;
    F.open_or_die(input, lcp, da, pos);
    pc.mark("open files");
    std::cout << "This is ebwt2clust. Input file: " << input << std::endl;

    // The EGSA goes through the GPU in chunks of SA positions (one chunk if it fits): the next chunk uploads while the current one is
    // scanned, the chunks are stitched like the ranks of a multi-GPU job, and the records stream out of pinned staging buffers
    // into <input>.clusters -- any size of input, O(chunk) memory on the device (e2s_ebwt2clust_chunked).
    std::string filename_out = input;
    filename_out.append(".clusters");
    const int ofd = ::open(filename_out.c_str(), O_WRONLY | O_CREAT | O_TRUNC, 0666);
    struct Out { int fd; bool bad; } sinkst{ofd, ofd < 0};
    auto sink = [](void* user, const void* data, size_t bytes) -> int {
        Out* o = (Out*)user;
        if (o->bad) return 0;                                   // (the reference's ofstream fails silently, too)
        const char* p = (const char*)data;
        while (bytes) {
            const ssize_t w = ::write(o->fd, p, bytes);
            if (w <= 0) { o->bad = true; return 0; }
            p += w; bytes -= (size_t)w;
        }
        return 0;
    };
    FileSource fs{&F, lcp, da, pos};
    const e2s_egsa_source src = fs.source();
    e2s_chunk_params cp{};
    cp.device = gpu; cp.k = k; cp.min_len = min_len; cp.lcp_bytes = lcp; cp.da_bytes = da; cp.pos_bytes = pos;
    cp.n_total = F.n; cp.chunk_entries = chunk_entries_env();
    e2s_job_report rep{};
    int rc = e2s_ebwt2clust_chunked(&cp, &src, sink, &sinkst, &rep);
    if (ofd >= 0) ::close(ofd);
    if (rc) {
        // (the job owns its handles: its message is this thread's handle-less one; no GPU shows up here, too)
        std::cerr << "cluster scan failed (" << rc << "): " << e2s_last_error(nullptr) << std::endl;
        if (ofd >= 0) ::unlink(filename_out.c_str());             // no partial .clusters left behind
        return 2;
    }
    if (pc.on) std::cerr << "[timing]   " << rep.n_chunks << " chunk(s), upload " << rep.seconds_upload * 1e3 << " ms of " << rep.seconds_total * 1e3 << " ms" << std::endl;
    pc.mark("HIP start-up + upload + scan + records -> .clusters");
    report.write("ebwt2clust", input, rep, F.n, F.n * (uint64_t)lcp, rep.n_records * 10ull);   // (E2S_CLI_JSON; the LCP column is what this tool uploads)
    const uint64_t n_closed = rep.n_closed;
    // the reference prints the count as unsigned int (ebwt2clust.cpp:88,137)
    std::cout << "Done. " << (unsigned int)n_closed << " clusters saved to output file." << std::endl;
    return 0;
}
