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
    EgsaFiles F;
    F.open_or_die(input, lcp, da, pos);
    pc.mark("open files");
    std::cout << "This is ebwt2clust. Input file: " << input << std::endl;

    e2s_handle h = nullptr;
    int rc = e2s_create(gpu, &h);
    if (rc) { std::cerr << "GPU unavailable: " << e2s_last_error(nullptr) << std::endl; return 2; }
    pc.mark("e2s_create (HIP start-up)");
    e2s_shard_desc d{};
    d.global_offset = 0; d.n_entries = F.n; d.n_total = F.n; d.halo_left = d.halo_right = 0;
    d.lcp_bytes = lcp; d.da_bytes = da; d.pos_bytes = pos;
    if (F.gesa) die_on(e2s_upload_gesa(h, &d, F.aos.p), h, "upload (.gesa)");
    else die_on(e2s_upload_lcp(h, &d, F.lcp.p, F.n ? F.bwt.p[F.n - 1] : 0), h, "upload (LCP)");   // cluster_lm only reads the LCP
    pc.mark("upload");
    die_on(e2s_cluster_scan(h, k, min_len, 0), h, "cluster scan");
    e2s_boundary b;
    die_on(e2s_boundary_desc(h, &b), h, "boundary");
    die_on(e2s_stitch(h, &b, 1, 0), h, "stitch");
    uint64_t n_closed = 0, n_rec = 0;
    die_on(e2s_cluster_counts(h, &n_closed, &n_rec), h, "counts");
    pc.mark("scan");
    // <input>.clusters: the packed records stream out of the library's pinned staging buffers into the file
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
    die_on(e2s_stream_clusters(h, sink, &sinkst, &n_rec), h, "fetch clusters");
    if (ofd >= 0) ::close(ofd);
    pc.mark("records -> .clusters");
    // the reference prints the count as unsigned int (ebwt2clust.cpp:88,137)
    std::cout << "Done. " << (unsigned int)n_closed << " clusters saved to output file." << std::endl;
    e2s_destroy(h);
    return 0;
}
