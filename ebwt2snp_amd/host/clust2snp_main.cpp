// clust2snp -- drop-in for the reference executable (clust2snp.cpp:969-1085): same getopt string (so
// -M and -b stay unparseable, H5), same defaults (0 = default), same files in and out, same phases.
// The per-cluster pass (find_events + find_variants) and the tail (consensus, distance, KisSNP2 text) run on the
// GPU through the C ABI; the histogram rule is host code in libe2s_hip.so, as is the tail for irregular inputs.
#include <unistd.h>

#include <cmath>
#include <ctime>
#include <fstream>
#include <vector>

#include "egsa_files.hpp"

static const int k_left_def = 31, k_right_def = 30, max_snvs_def = 3, mcov_out_def = 5, max_gap_def = 10;
static const int consensus_reads_def = 20, max_err_def = 2, lcp_def = 1, da_def = 4, pos_def = 1;
static const double pval_def = 0.99;

static void help() {
    std::cout << "clust2snp [options]\nOptions:\n"
                 "-h          Print this help.\n"
                 "-i <arg>    Input fasta file containing the samples' reads (REQUIRED).\n"
                 "-n <arg>    Number of reads in the first sample (REQUIRED).\n"
                 "-L <arg>    Length of left-context, SNP included (default: " << k_left_def << ").\n"
                 "-R <arg>    Length of right context, SNP excluded (default: " << k_right_def << ").\n"
                 "-g <arg>    Maximum allowed gap length in indel (default: " << max_gap_def << "). If 0, indels are disabled.\n"
                 "-v <arg>    Maximum number of non-isolated SNPs in left-contexts. The central SNP/indel is excluded from this count (default: " << max_snvs_def << ").\n"
                 "-c <arg>    Extract this maximum number of reads per individual to compute consensus of left-context (default: " << consensus_reads_def << ").\n"
                 "-e <arg>    Mismatches allowed between DNA fragments forming consensus of left-context (default: " << max_err_def << ").\n"
                 "-m <arg>    Minimum cluster length per individual (default: " << mcov_out_def << "). The minimum cluster length (for the 2 individuals) is 2*<arg>.\n"
                 "-p <arg>    Automatically choose max cluster length so that this fraction of bases is analyzed (default: \n"
                 "            " << pval_def << "). In any case, the maximum cluster length will not exceed the value specified with -M.\n"
                 "-M <arg>    Maximum cluster length. Read the description of option -p.\n"
                 "-x <arg>    Byte size of LCP integers in input EGSA/BCR file (default: " << lcp_def << ").\n"
                 "-y <arg>    Byte size of DA integers (read number) in input EGSA/BCR file (default: " << da_def << ").\n"
                 "-z <arg>    Byte size of pos integers (position in read) in input EGSA/BCR file (default: " << pos_def << ").\n\n"
                 "\nNeeds the EGSA of the input (see ebwt2clust -h) and <input>.clusters written by ebwt2clust.\n"
                 "Output: SNPs/indels in KisSNP2 format (a fasta file) in <input without .fast*>.snp; many events are\n"
                 "reported twice, once per strand.\n";
    exit(0);
}

int main(int argc, char** argv) {
    srand(time(NULL));
    if (argc < 3) help();
    int k_left = 0, k_right = 0, max_snvs = 0, mcov_out = 0, max_gap = 0, consensus_reads = 0, max_err = 0;
    int lcp = 0, da = 0, pos = 0;
    double pval = 0;
    uint64_t nr_reads1 = 0;
    std::string input;
    int opt;
    while ((opt = getopt(argc, argv, "hi:n:p:v:L:R:m:g:c:x:y:z:e:")) != -1) {
        switch (opt) {
            case 'h': help(); break;
            case 'i': input = std::string(optarg); break;
            case 'n': nr_reads1 = atoi(optarg); break;
            case 'm': mcov_out = atoi(optarg); break;
            case 'g': max_gap = atoi(optarg); break;
            case 'L': k_left = atoi(optarg); break;
            case 'c': consensus_reads = atoi(optarg); break;
            case 'R': k_right = atoi(optarg); break;
            case 'p': pval = atof(optarg); break;
            case 'v': max_snvs = atoi(optarg); break;
            case 'e': max_err = atof(optarg); break;
            case 'x': lcp = atoi(optarg); break;
            case 'y': da = atoi(optarg); break;
            case 'z': pos = atoi(optarg); break;
            default: help(); return -1;
        }
    }
    lcp = lcp == 0 ? lcp_def : lcp;
    da = da == 0 ? da_def : da;
    pos = pos == 0 ? pos_def : pos;
    max_err = max_err == 0 ? max_err_def : max_err;
    consensus_reads = consensus_reads == 0 ? consensus_reads_def : consensus_reads;
    max_gap = max_gap == 0 ? max_gap_def : max_gap;
    k_left = k_left == 0 ? k_left_def : k_left;
    k_right = k_right == 0 ? k_right_def : k_right;
    pval = pval == 0 ? pval_def : pval;
    max_snvs = max_snvs == 0 ? max_snvs_def : max_snvs;      // parsed, never used (clust2snp.cpp:648)
    mcov_out = mcov_out == 0 ? mcov_out_def : mcov_out;
    (void)max_snvs;
    if (input.compare("") == 0 || nr_reads1 == 0) help();

    PhaseClock pc;
    CliReport report;
    EgsaFiles F;
    F.open_or_die(input, lcp, da, pos);
    std::cout << "This is clust2snp." << std::endl
              << "Input file: " << input << std::endl
              << "Left-extending GSA ranges by " << k_left << " bases." << std::endl
              << "Right context length: at most " << k_right << " bases." << std::endl;
    std::string clusters_path = input;
    clusters_path.append(".clusters");
    MappedFile CL;
    if (!CL.open(clusters_path)) {
        std::cout << "\nERROR: Could not find BWT clusters file \"" << clusters_path << "\"" << std::endl << std::endl;
        help();
    }
    std::string filename_out = input;
    filename_out = filename_out.substr(0, filename_out.rfind(".fast"));
    filename_out.append(".snp");
    std::cout << "Output events will be stored in " << filename_out << std::endl;

    pc.mark("open files");
    int rc;

    // statistics(), clust2snp.cpp:877-966: one pass over the records of the file (host threads)
    e2s_hist hs;
    rc = e2s_hist_of_records(CL.p, CL.n / 10, &hs);
    if (rc == E2S_ERR_INPUT) { std::cerr << "load .clusters failed (" << rc << "): .clusters records must be ascending and non-overlapping (H16)" << std::endl; return 2; }
    if (rc) { std::cerr << "histogram failed (" << rc << ")" << std::endl; return 2; }
    int max_clust_length = 0;
    uint64_t n_clust = 0, n_bases = 0;
    rc = e2s_auto_max_clust_length(&hs, 1, mcov_out, pval, &max_clust_length, &n_clust, &n_bases);
    if (rc == E2S_ERR_REF_CRASH) {
        std::cerr << "no usable cluster record: the reference divides by zero here (SIGFPE); nothing written" << std::endl;
        return 136;
    }
    if (rc) { std::cerr << "statistics failed (" << rc << ")" << std::endl; return 2; }
    {
        // the histogram print-out of clust2snp.cpp:916-934 (with the double-counted last record, H4)
        uint64_t freq[E2S_MAX_CLUST_LEN + 1];
        for (int i = 0; i <= E2S_MAX_CLUST_LEN; ++i) freq[i] = hs.freq[i];
        if (hs.last_len <= E2S_MAX_CLUST_LEN) freq[hs.last_len]++;
        uint64_t max = 0, max_len = 0, cumulative = 0;
        for (int i = 1; i <= E2S_MAX_CLUST_LEN; ++i) { if (freq[i] * i > max) max = freq[i] * i; }
        for (int i = 0; i <= E2S_MAX_CLUST_LEN; ++i) if (freq[i]) max_len = i;
        std::cout << "\nDistribution of base coverage: " << std::endl;
        std::cout << "\ncluster length\t# bases in a cluster with this length\t cumulative fraction (from 2m = " << 2 * mcov_out << ")" << std::endl;
        for (uint64_t i = 0; i <= max_len; ++i) {
            std::cout << i << "\t";
            for (uint64_t j = 0; j < (100 * freq[i] * i) / max; ++j) std::cout << "-";
            std::cout << "\t" << freq[i] * i;
            if ((int64_t)i >= 2 * mcov_out) { cumulative += freq[i] * i; std::cout << "\t" << double(cumulative) / double(n_bases); }
            std::cout << std::endl;
        }
        std::cout << "\nCluster sizes allowed: [" << mcov_out * 2 << "," << max_clust_length << "]" << std::endl;
    }

    std::cout << "(1/4) Filtering relevant clusters ... " << std::endl;
    pc.mark("statistics");
    // find_events + extract_variants + to_file (clust2snp.cpp:788-868): the EGSA goes through the GPU in chunks of SA positions (one
    // chunk if it fits), the next one uploading while the current one is processed; the reads are resident in HBM throughout and
    // consensus, distance and the KisSNP2 text run there too (the host tail takes over for an irregular FASTA or -L outside 1..64)
    FileSource fs{&F, lcp, da, pos};
    const e2s_egsa_source src = fs.source();
    e2s_chunk_params cp{};
    cp.device = 0; cp.k = 0; cp.min_len = 0; cp.lcp_bytes = lcp; cp.da_bytes = da; cp.pos_bytes = pos;
    cp.n_total = F.n; cp.chunk_entries = chunk_entries_env();
    e2s_c2s_options co{};
    co.nr_reads1 = nr_reads1; co.k_left = k_left; co.k_right = k_right; co.mcov_out = mcov_out; co.consensus_reads = consensus_reads;
    co.max_gap = max_gap; co.max_err = max_err; co.pval = pval;
    e2s_job_report rep{};
    rc = e2s_clust2snp_chunked(&cp, &co, max_clust_length, &src, CL.p, CL.n / 10, input.c_str(), filename_out.c_str(), &rep);
    if (pc.on) std::cerr << "[timing]   " << rep.n_chunks << " chunk(s), upload " << rep.seconds_upload * 1e3 << " ms of " << rep.seconds_total * 1e3 << " ms" << std::endl;
    pc.mark("HIP start-up + upload + candidates + tail + .snp");
    std::cout << " 100% done." << std::endl;
    std::cout << "Done. " << rep.n_candidates << " potential variants detected (some might be detected twice: on fw and rev strands)" << std::endl;
    std::cout << "(2/4) Extracting reads from fasta file ..." << std::endl;
    std::cout << "(3/4) Filtering " << rep.n_candidates << " candidates and computing consensus of left-contexts ... " << std::endl;
    std::cout << "(4/4) Computing edit distances and saving SNPs/indels to file ... " << std::endl;
    if (rc == E2S_ERR_REF_CRASH) {
        std::cerr << "no candidate variant: the reference crashes here (SIGSEGV, clust2snp.cpp:531); no .snp written" << std::endl;
        return 139;
    }
    if (rc) { std::cerr << "clust2snp failed (" << rc << "): " << e2s_last_error(nullptr) << std::endl; return 2; }
    {
        e2s_job_report r2 = rep;                                  // (the statistics ran here, on the host, before the job)
        r2.max_clust_length = max_clust_length; r2.n_records = CL.n / 10;
        report.write("clust2snp", input, r2, F.n, F.n * (uint64_t)(lcp + 1), rep.snp_bytes);
    }
    std::cout << "Done. " << std::endl;
    return 0;
}
