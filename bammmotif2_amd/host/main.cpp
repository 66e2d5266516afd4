// BaMMmotif OUTDIR FASTA [options] -- MI355X drop-in for the reference driver
// (/root/reference/src/refinement/mainBaMM.cpp, Global.cpp).  The EM itself runs on the GPU
// through the C ABI (include/bamm_em.h); everything here is host plumbing with the reference's
// flags, defaults, messages and output files, including --scoreSeqset (.occurrence), --FDR
// (cross-validated .zoops.stats), --saveLogOdds and --advanceEM (EM::mask); --gpus N shards the
// sequences (--EM) and spreads the cross-validation folds (--FDR) over N GPUs.  Not ported (exit
// with a clear message): --CGS, non-STANDARD alphabets.
#include <omp.h>

#include <algorithm>
#include <memory>
#include <thread>
#include <sys/stat.h>
#include <unistd.h>

#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <iomanip>
#include <iostream>
#include <limits>
#include <map>
#include <set>

#include "bamm_host.h"

using namespace bammhost;

namespace {

// Threads beside main(): exit() runs the static destructors of the HIP runtime, the OpenMP runtime and this program under
// whatever is still running, so die() -- called on the main thread only; the side threads report through strings -- joins
// every one of them first.  All three do bounded work (no collective: the sharded ranks are joined where they start).
std::thread g_hip_warmup;                  // brings the HIP runtime up while the FASTA file is read (main)
std::thread g_neg_thread;                  // samples, packs and uploads the negatives beside the main run
std::thread g_fold_thread;                 // overlap mode: a motif's folds train while its main run does

[[noreturn]] void die(const std::string& msg) {
    std::cerr << msg << std::endl;
    for (std::thread* t : {&g_hip_warmup, &g_neg_thread, &g_fold_thread})
        if (t->joinable() && t->get_id() != std::this_thread::get_id()) t->join();
    exit(1);
}

[[noreturn]] void die_abi(const char* what) { die(std::string("Error: ") + what + ": " + bamm_last_error()); }

void print_help() {
    printf("\n==================================================================\n");
    printf("\n SYNOPSIS:  BaMMmotif OUTDIR SEQFILE [options] \n\n");
    printf("\t DESCRIPTION \n");
    printf("\t\t Learn Bayesian inhomogeneous Markov models (BaMMs) from sequence data (EM on an MI355X GPU).\n\n");
    printf("\t OUTDIR:  output directory for all results. \n");
    printf("\t SEQFILE: file with sequences from positive set in FASTA format\n\n");
    printf("\t OPTIONS (same names and defaults as the reference, Global.cpp:142-341):\n");
    printf("\t\t --basename <STRING> --negSeqFile <FILE> --ss --alphabet STANDARD\n");
    printf("\t\t --bindingSiteFile <FILE> | --PWMFile <FILE> | --BaMMFile <FILE>   --maxPWM <INT>\n");
    printf("\t\t -k, --order <INT> (2)   -a, --alpha <FLOAT>..   -b, --beta <FLOAT> (7)   -r, --gamma <FLOAT> (3)\n");
    printf("\t\t --extend <INT> [<INT>]   --bgModelFile <FILE>   -K, --Order <INT> (2)   -A, --Alpha <FLOAT>..\n");
    printf("\t\t --EM   -q <FLOAT> (0.3)   --optimizeQ   --verbose   --saveBaMMs   --saveInitialBaMMs\n");
    printf("\t EXTENSIONS of this build:\n");
    printf("\t\t --maxEMIterations <INT> (1000)   -e, --epsilon <FLOAT> (0.01)   --device <INT> (0)\n");
    printf("\t\t --timing (wall time per stage on stderr)   --hostSeeding (initFromPWM's pass on the host)\n");
    printf("\t\t --hostPacking (Sequence.cpp's encoding and the background counts on the host instead of the device)\n");
    printf("\t\t --hostSampler (SeqGenerator's negative sampler on the host instead of the device)\n");
    printf("\t\t --gpus <INT> (1)   --deviceList <INT,INT,..>\n");
    printf("\t\t\t --EM: the sequences are sharded over the GPUs, one RCCL all-reduce of the count table per iteration;\n");
    printf("\t\t\t --FDR: cross-validation fold f runs on GPU f mod N (FDR.cpp:37 runs the folds on host threads).\n");
    printf("\t\t\t Output files do not depend on the number of GPUs.\n");
    printf("\n==================================================================\n");
}

// Tokeniser in the spirit of getopt_pp (src/getopt_pp/getopt_pp.cpp:71-141): "--long", "-s", combined
// short flags, values = following tokens that do not look like options (negative numbers do not).
struct Args {
    std::map<std::string, std::vector<std::string>> longs;
    std::map<char, std::vector<std::string>> shorts;
    std::set<std::string> used_long;
    std::set<char> used_short;

    static bool looks_like_option(const std::string& t) {
        if (t.size() < 2 || t[0] != '-') return false;
        if (isdigit((unsigned char)t[1]) || t[1] == '.') return false;      // -3, -.5 are values
        return true;
    }
    Args(int n, char** v) {
        std::vector<std::string>* cur = nullptr;
        for (int i = 1; i < n; i++) {
            std::string t = v[i];
            if (looks_like_option(t)) {
                if (t[1] == '-') {
                    cur = &longs[t.substr(2)];
                } else {
                    for (size_t c = 1; c < t.size(); c++) cur = &shorts[t[c]];
                }
            } else if (cur) {
                cur->push_back(t);
            }
        }
    }
    bool present(char s, const std::string& l) {
        bool p = false;
        if (s && shorts.count(s)) { used_short.insert(s); p = true; }
        if (!l.empty() && longs.count(l)) { used_long.insert(l); p = true; }
        return p;
    }
    const std::vector<std::string>* values(char s, const std::string& l) {
        if (s && shorts.count(s)) { used_short.insert(s); return &shorts[s]; }
        if (!l.empty() && longs.count(l)) { used_long.insert(l); return &longs[l]; }
        return nullptr;
    }
    template <class T>
    bool get(char s, const std::string& l, T& out) {
        const auto* v = values(s, l);
        if (!v || v->empty()) return false;
        std::stringstream ss((*v)[0]);
        T tmp;
        if (!(ss >> tmp)) die("Error: bad value for option " + (l.empty() ? std::string(1, s) : l));
        out = tmp;
        return true;
    }
    bool get_str(char s, const std::string& l, std::string& out) {
        const auto* v = values(s, l);
        if (!v || v->empty()) return false;
        out = (*v)[0];
        return true;
    }
    template <class T>
    bool get_vec(char s, const std::string& l, std::vector<T>& out) {
        const auto* v = values(s, l);
        if (!v) return false;
        for (const auto& t : *v) { std::stringstream ss(t); T x; if (ss >> x) out.push_back(x); }
        return true;
    }
    bool remain() const {
        for (auto& kv : longs) if (!used_long.count(kv.first)) return true;
        for (auto& kv : shorts) if (!used_short.count(kv.first)) return true;
        return false;
    }
};

struct Options {                       // Global.cpp:6-96 defaults
    std::string out_dir, fasta, basename, neg_fasta, alphabet = "STANDARD";
    std::string seed_file, seed_tag, bg_file;
    bool ss = false, EM = false, CGS = false, FDR = false, score = false, verbose = false;
    bool optimizeQ = false, advanceEM = false, saveBaMMs = true, saveInitial = false, mops = false, zoops = true;
    bool genericNeg = false, savePRs = true, savePvalues = false, saveLogOdds = false;
    float pvalCutoff = 0.0001f;
    size_t maxPWM = std::numeric_limits<size_t>::max();
    uint32_t K = 2, Kbg = 2;
    std::vector<float> alpha{1.f, 1.f, 1.f}, alpha_bg{1.f, 1.f, 1.f};
    float beta = 7.0f, gamma = 3.0f, q = 0.3f, f = 0.05f, epsilon = 0.01f;
    std::vector<size_t> extend{0, 0};
    size_t cvFold = 4, mFold = 1, sOrder = 2, threads = 4;
    uint32_t max_iter = 1000;
    int device = 0;
    bool timing = false, hostSeeding = false, hostPacking = false, hostSampler = false, forceComm = false, debug = false;
    size_t gpus = 1;                   // --gpus N: devices device .. device+N-1 (or --deviceList)
    std::vector<int> device_list;
};

template <class T>
void fit(std::vector<T>& v, size_t n) {     // Global.cpp:210-223: truncate or pad with the last value
    if (v.size() > n) v.resize(n);
    else if (v.size() < n) v.resize(n, v.empty() ? T(1) : v.back());
}

Options parse(int nargs, char** args) {
    if (nargs < 3) {
        std::cerr << "Error: Arguments are missing! \n" << std::endl;
        print_help();
        exit(1);
    }
    Options o;
    o.out_dir = args[1];
    struct stat st;
    if (stat(o.out_dir.c_str(), &st) != 0) {                 // utils.h:154-165
        std::cout << "New output directory is created automatically.\n";
        if (system(("mkdir -p " + o.out_dir).c_str()) != 0) {
            std::cerr << "Error: Directory " << o.out_dir << " could not be created." << std::endl;
            exit(-1);
        }
    }
    o.fasta = args[2];
    Args a(nargs - 2, args + 2);                              // the FASTA path plays argv[0] (Global.cpp:142)
    if (a.present('h', "help")) { print_help(); exit(1); }
    if (!a.get_str(0, "basename", o.basename)) o.basename = base_name(o.fasta);
    a.present(0, "maskPosSequenceSet");
    if (!a.get_str(0, "negSeqFile", o.neg_fasta)) o.neg_fasta = o.fasta;
    o.genericNeg = a.present(0, "genericNeg");
    a.get_str(0, "alphabet", o.alphabet);
    o.ss = a.present(0, "ss");
    { std::string tmp; a.get_str(0, "intensityFile", tmp); }
    if (a.get_str(0, "bindingSiteFile", o.seed_file)) o.seed_tag = "bindingsites";
    else if (a.get_str(0, "PWMFile", o.seed_file)) o.seed_tag = "PWM";
    else if (a.get_str(0, "BaMMFile", o.seed_file)) o.seed_tag = "BaMM";
    else { fprintf(stderr, "Error: No initial model is provided.\n"); exit(1); }
    a.get(0, "maxPWM", o.maxPWM);
    o.mops = a.present(0, "mops");
    a.get(0, "zoops", o.zoops);
    a.get('k', "order", o.K);
    if (a.present('a', "alpha")) {
        o.alpha.clear();
        a.get_vec('a', "alpha", o.alpha);
        fit(o.alpha, o.K + 1);
    } else {
        fit(o.alpha, o.K + 1);
        a.get('b', "beta", o.beta);
        a.get('r', "gamma", o.gamma);
        for (uint32_t k = 1; k <= o.K; k++) o.alpha[k] = o.beta * powf(o.gamma, (float)k);   // Global.cpp:227-232
    }
    if (a.present(0, "extend")) {
        o.extend.clear();
        a.get_vec(0, "extend", o.extend);
        if (o.extend.size() < 1 || o.extend.size() > 2) { fprintf(stderr, "--extend format error.\n"); exit(1); }
        if (o.extend.size() == 1) o.extend.resize(2, o.extend.back());
    }
    a.get_str(0, "bgModelFile", o.bg_file);
    a.get('K', "Order", o.Kbg);
    if (a.present('A', "Alpha")) {
        o.alpha_bg.clear();
        a.get_vec('A', "Alpha", o.alpha_bg);
        fit(o.alpha_bg, o.Kbg + 1);
    } else {
        fit(o.alpha_bg, o.Kbg + 1);
        for (uint32_t k = 1; k <= o.Kbg; k++) o.alpha_bg[k] = 10.0f;                           // Global.cpp:274-278
    }
    o.EM = a.present(0, "EM");
    if ((o.CGS = a.present(0, "CGS"))) {
        for (const char* n : {"noInitialZ", "noAlphaOpti", "GibbsMH", "dissample", "noZSampling", "noQSampling"}) a.present(0, n);
    }
    a.present(0, "debugAlphas");
    a.present(0, "generatePseudoSet");
    a.get('q', "", o.q);
    a.get('f', "", o.f);
    if ((o.FDR = a.present(0, "FDR"))) {
        a.get('m', "mFold", o.mFold);
        a.get('n', "cvFold", o.cvFold);
        a.get('s', "sOrder", o.sOrder);
    }
    o.score = a.present(0, "scoreSeqset");
    a.get(0, "pvalCutoff", o.pvalCutoff);
    o.verbose = a.present(0, "verbose");
    o.debug = a.present(0, "debug");
    o.saveBaMMs = a.present(0, "saveBaMMs");                  // presence overwrites the default (getopt_pp.h:497)
    o.saveInitial = a.present(0, "saveInitialBaMMs");
    a.get(0, "savePRs", o.savePRs);
    o.savePvalues = a.present(0, "savePvalues");
    o.saveLogOdds = a.present(0, "saveLogOdds");
    for (const char* n : {"saveBgModel", "makeMovie", "B2", "B3", "B3prime"}) a.present(0, n);
    o.optimizeQ = a.present(0, "optimizeQ");
    o.advanceEM = a.present(0, "advanceEM");
    a.get(0, "threads", o.threads);
    omp_set_num_threads((int)std::max<size_t>(1, o.threads));   // Global.cpp:331-333 (default 4)
    // packing, the negative sampler and the sorts give the same bytes however they are cut: all granted cores
    bamm_set_host_threads((uint32_t)std::max<size_t>(o.threads, (size_t)host_parallelism()));
    // extensions of this build (the reference advertises but never parses the first two, Global.cpp:479-491)
    a.get(0, "maxEMIterations", o.max_iter);
    a.get('e', "epsilon", o.epsilon);
    a.get(0, "device", o.device);
    o.timing = a.present(0, "timing");
    o.hostSeeding = a.present(0, "hostSeeding");
    o.hostPacking = a.present(0, "hostPacking");
    o.hostSampler = a.present(0, "hostSampler");
    a.get(0, "gpus", o.gpus);
    {   // --deviceList 0,1,2: explicit devices (a device may appear twice for the fold replicas of --FDR; the
        // sharded --EM wants distinct ones, RCCL has one rank per GPU)
        std::string list;
        if (a.get_str(0, "deviceList", list)) {
            std::stringstream ss(list);
            std::string tok;
            while (std::getline(ss, tok, ',')) if (!tok.empty()) o.device_list.push_back(atoi(tok.c_str()));
            if (o.device_list.empty()) { fprintf(stderr, "--deviceList format error.\n"); exit(1); }
            o.gpus = o.device_list.size();
        }
    }
    if (o.gpus < 1) o.gpus = 1;
    if (o.device_list.empty()) for (size_t d = 0; d < o.gpus; d++) o.device_list.push_back(o.device + (int)d);
    o.forceComm = a.present(0, "forceComm");
    if (a.remain()) {
        print_help();
        std::cerr << "Oops! Unknown option(s) remaining... \n\n";
        exit(1);
    }
    return o;
}

}  // namespace

int main(int nargs, char* args[]) {
    auto t0_wall = std::chrono::high_resolution_clock::now();
    // --timing: wall time per stage on stderr (stdout stays the reference's)
    bool timing = false;
    auto t_stage = t0_wall;
    auto stage = [&](const char* what) {
        if (!timing) return;
        auto now = std::chrono::high_resolution_clock::now();
        std::cerr << "[timing] " << what << ": " << std::chrono::duration<double>(now - t_stage).count() << " s" << std::endl;
        t_stage = now;
    };
    std::cout << std::endl
              << "======================================" << std::endl
              << "=      Welcome to use BaMM!motif     =" << std::endl
              << "=                   Version 2.0      =" << std::endl
              << "=     MI355X build (bammmotif2_amd)  =" << std::endl
              << "======================================" << std::endl;
    srand(42);                                               // mainBaMM.cpp:22
    Options o = parse(nargs, args);
    timing = o.timing;
    auto epoch = [] { return std::chrono::duration<double>(std::chrono::system_clock::now().time_since_epoch()).count(); };
    if (timing) fprintf(stderr, "[timing-abs] main entered at %.4f\n", epoch() - std::chrono::duration<double>(std::chrono::high_resolution_clock::now() - t0_wall).count());
    if (o.alphabet != "STANDARD") die("Error: this build supports --alphabet STANDARD only.");
    if (o.CGS) die("Error: --CGS (collapsed Gibbs sampling) is not part of the MI355X build.");
    if (o.K > BAMM_MAX_ORDER) die("Error: model order above 10 is not supported (kmer_ spans 11 bases).");

    // the HIP runtime takes 0.1-0.2 s to come up on first use: it does so on a thread of its own while the FASTA
    // file is read and packed (nothing is decided there: the contexts proper are created where they always were)
    // ... and since the packing itself runs on the device, the first slot's CONTEXT is created there as well (the
    // runtime's first use of a device, its stream, the library's code objects): ready when the FASTA file is
    static bamm_ctx* warm_ctx = nullptr;                     // written by the warm-up thread only; read after it was joined
    static int warm_device = 0;
    warm_device = o.device_list.empty() ? 0 : o.device_list[0];
    std::thread& hip_warmup = g_hip_warmup;
    if (o.EM || o.score || o.FDR) hip_warmup = std::thread([] {
        int n = 0;
        if (bamm_device_count(&n) == BAMM_OK && bamm_ctx_create(warm_device, nullptr, &warm_ctx) != BAMM_OK) warm_ctx = nullptr;
    });
    struct Joiner { std::thread& t; ~Joiner() { if (t.joinable()) t.join(); } } hip_warmup_joiner{hip_warmup};

    std::string err;
    FastaSet pos;
    if (read_fasta(o.fasta, pos, err)) die(err);
    if (pos.size() < o.cvFold) die("Error: Input sequences are too few for training! \n");
    stage("read FASTA");
    const bool need_gpu = o.EM || o.score || o.FDR;
    // one slot per GPU (a single one unless --gpus / --deviceList): context, resident sets, RCCL rank
    struct Dev {
        int device = 0;
        bamm_ctx* ctx = nullptr;
        bamm_seqs* full = nullptr;         // every kept positive (scoring, fold replicas, single-GPU EM)
        bamm_seqs* shard = nullptr;        // this GPU's range of the kept positives (the full set with one GPU)
        bamm_seqs* neg = nullptr;          // the sampled negatives, all of them (--scoreSeqset scores them on the first GPU)
        bamm_seqs* neg_cv = nullptr;       // every cvFold-th negative: all the folds of --FDR ever score (FDR.cpp:58-60)
        bamm_comm* comm = nullptr;
        uint64_t begin = 0, end = 0;
    };
    const size_t ndev = need_gpu ? o.device_list.size() : 1;
    std::vector<Dev> devs(ndev);
    for (size_t d = 0; d < ndev; d++) devs[d].device = o.device_list[d];
    auto make_ctx = [&](Dev& dv) {
        if (hip_warmup.joinable()) hip_warmup.join();
        if (!dv.ctx && warm_ctx && dv.device == warm_device) { dv.ctx = warm_ctx; warm_ctx = nullptr; }   // the one the warm-up made
        if (!dv.ctx && bamm_ctx_create(dv.device, nullptr, &dv.ctx)) die_abi("no usable MI355X");
    };
    bamm_packed* packed = nullptr;
    bamm_seqs* dseqs_all = nullptr;                          // every positive record, resident (packing, seeding, then EM)
    // the stream stands at srand(42) (above; nothing between draws from it): the N draws are taken on all host threads
    if (need_gpu && !o.hostPacking) {
        // Sequence::Sequence where the data will live (csrc/prep.hip): the same packed set, and the resident set with it
        make_ctx(devs[0]);
        if (bamm_seqs_from_codes(devs[0].ctx, pos.codes.data(), pos.off.data(), pos.size(), o.ss ? 1 : 0, 42u, &packed, &dseqs_all)) die_abi("packing sequences");
        stage("encode + 2-bit pack on the device, resident set (Sequence.cpp incl. rand() protocol)");
    } else {
        if (bamm_pack_codes_seeded(pos.codes.data(), pos.off.data(), pos.size(), o.ss ? 1 : 0, 42u, &packed)) die_abi("packing sequences");
        stage("encode + 2-bit pack (Sequence.cpp incl. rand() protocol)");
    }
    // (records beyond 8192 positions leave the register-resident kernels for the window-by-window path, csrc/long_seq.hip;
    // initFromPWM's pass and EM::mask keep their per-wave arrays in a global scratch region there: no limit on the length)

    if (o.verbose) std::cout << std::endl << "************************" << std::endl << "*   Background Model   *" << std::endl << "************************" << std::endl;
    BgModel bg;
    if (o.bg_file.empty()) {
        if (dseqs_all) {                                     // the counting pass over the resident set (BackgroundModel.cpp:26-42)
            bg.K = o.Kbg; bg.alpha = o.alpha_bg; bg.v.assign(bamm_bg_size(o.Kbg), 0.f);
            if (bamm_seqs_bg_model(devs[0].ctx, dseqs_all, o.Kbg, o.alpha_bg.data(), bg.v.data())) die_abi("background model");
        } else if (bg_learn(packed, o.Kbg, o.alpha_bg, bg)) die_abi("background model");
    } else if (bg_read(o.bg_file, bg, err)) {
        die(err);
    }
    if (bg_write(o.out_dir, o.basename, bg, err)) die(err);   // always saved (mainBaMM.cpp:51)
    stage("background model");

    if (o.verbose) std::cout << std::endl << "***************************" << std::endl << "*   Initial Motif Model   *" << std::endl << "***************************" << std::endl;
    std::vector<uint64_t> off(pos.size() + 1, 0);
    for (size_t n = 0; n < pos.size(); n++) off[n + 1] = off[n] + packed->len[n];
    SeedDevice seed_dev;
    std::vector<uint32_t> yK;
    if (need_gpu && o.seed_tag == "PWM" && !o.hostSeeding) {
        // Motif::initFromPWM's pass over the sequences runs on the device: upload first
        make_ctx(devs[0]);
        if (!dseqs_all && bamm_seqs_upload(devs[0].ctx, packed, 0, packed->n_seqs, &dseqs_all)) die_abi("upload");
        seed_dev.ctx = devs[0].ctx; seed_dev.seqs = dseqs_all;
        stage("device context + upload of the positives");
    } else if (o.seed_tag == "PWM") {
        yK.resize(packed->total_len ? packed->total_len : 1);
        if (bamm_unpack_y(packed, o.K, yK.data())) die_abi("unpack");
    }
    SeedSet seeds;
    // MotifSet hands Global::bgModelOrder and the model's v to every Motif (mainBaMM.cpp:60-70)
    if (load_seeds(o.seed_file, o.seed_tag, (uint32_t)o.extend[0], (uint32_t)o.extend[1], o.K, o.alpha, o.maxPWM, o.q, bg,
                   yK.empty() ? nullptr : yK.data(), off.data(), pos.size(), seeds, err, seed_dev.ctx ? &seed_dev : nullptr)) die(err);
    stage("seed models (initFromPWM / BaMM / sites)");

    // drop sequences shorter than the widest motif (mainBaMM.cpp:75-83)
    std::vector<uint8_t> keep(pos.size(), 1);
    size_t posN = 0;
    for (size_t n = 0; n < pos.size(); n++) { keep[n] = packed->len[n] >= seeds.max_w; posN += keep[n]; }
    if (posN < o.cvFold) { std::cerr << "There are " << posN << " sequences longer than input motif. Exit!\n"; exit(1); }

    if (o.verbose) std::cout << std::endl << "*********************" << std::endl << "*   BaMM Training   *" << std::endl << "*********************" << std::endl;
    ByteVec neg_codes;
    size_t negN = 0;                        // negatives the reference would hold (all of them, sampled or not)
    std::vector<uint32_t> neg_cv_len;       // lengths of the folds' subset (every cvFold-th negative)
    std::vector<uint64_t> neg_off{0};
    std::vector<uint32_t> neg_len;          // all negatives: only sampled for --scoreSeqset
    std::thread& neg_thread = g_neg_thread; // samples, packs and uploads the negatives beside the main run (joined by die() too)
    struct NegJoin { std::thread& t; ~NegJoin() { if (t.joinable()) t.join(); } } neg_join{neg_thread};
    std::string neg_err;
    bool neg_on_device = false;
    double neg_t_sample = 0, neg_t_pack = 0;
    std::vector<uint32_t> kept_len;
    // The plan: which GPU slot does what (FDR.cpp:37-127, mainBaMM.cpp:131-147).
    //   * the main EM run is sharded over its group of slots, one all-reduce of the count table per iteration: RCCL when
    //     the group's devices are distinct, else (a device listed twice: self-tests on a 1-GPU box) the host-staged sum;
    //   * --FDR trains fold f on fold_slot[f] from the SEED model, so the folds do not wait for the main run: with at
    //     least cvFold + 1 slots the folds take the last cvFold of them and the main run the others, AT THE SAME TIME
    //     (8 GPUs, 5 folds: 3 + 5, no idle device); with fewer slots the main run uses all of them first and the folds
    //     then go round them (fold f on slot f mod N).
    //   --advanceEM --optimizeQ re-estimates q after every sequence (EM.cpp:321): that chain runs on one slot.
    const size_t cvF = std::max<size_t>(1, o.cvFold);
    const bool overlap = need_gpu && o.FDR && o.EM && ndev >= cvF + 1;
    std::vector<size_t> em_slots, fold_slot(cvF, 0);
    for (size_t d = 0; d < (overlap ? ndev - cvF : ndev); d++) em_slots.push_back(d);
    for (size_t f = 0; f < cvF; f++) fold_slot[f] = overlap ? ndev - cvF + f : f % ndev;
    if (o.advanceEM && o.optimizeQ) em_slots.resize(1);
    const size_t ne = em_slots.size();
    std::set<int> em_devices;
    for (size_t d : em_slots) em_devices.insert(o.device_list[d]);
    const bool distinct = em_devices.size() == ne;
    const bool sharded = ne > 1 && o.EM;
    auto in_em_group = [&](size_t d) { return d < ne; };
    auto runs_folds = [&](size_t d) { return o.FDR && std::find(fold_slot.begin(), fold_slot.end(), d) != fold_slot.end(); };
    if (timing && need_gpu) {
        std::cerr << "  plan over " << ndev << " GPU slot(s) [devices";
        for (int dv : o.device_list) std::cerr << ' ' << dv;
        std::cerr << "]:";
        if (o.EM) std::cerr << " main EM on slot(s) 0.." << ne - 1 << (sharded ? (distinct ? " (sharded, RCCL all-reduce per iteration)" : " (sharded, host-staged all-reduce: a device is listed twice)") : "");
        if (o.FDR) {
            std::cerr << "; fold -> slot";
            for (size_t f = 0; f < cvF; f++) std::cerr << ' ' << f << "->" << fold_slot[f];
            std::cerr << (overlap ? " (while the main run trains)" : " (after the main run)");
        }
        if (o.score) std::cerr << "; --scoreSeqset on slot 0";
        std::cerr << std::endl;
    }
    if (need_gpu) {
        for (auto& dv : devs) make_ctx(dv);
        bamm_packed* use = packed;
        bamm_packed* filtered = nullptr;
        if (posN != pos.size()) {                            // re-pack only the kept records; kmers are position-local
            std::vector<uint64_t> kept_off{0};
            std::vector<uint64_t> km;
            std::vector<uint32_t> y10(packed->total_len);
            bamm_unpack_y(packed, BAMM_MAX_ORDER, y10.data());
            for (size_t n = 0; n < pos.size(); n++)
                if (keep[n]) { for (uint64_t i = off[n]; i < off[n + 1]; i++) km.push_back(y10[i]); kept_off.push_back(km.size()); }
            if (bamm_pack_kmers(km.data(), kept_off.data(), kept_off.size() - 1, &filtered)) die_abi("re-pack");
            use = filtered;
        }
        if (dseqs_all && use != packed) { bamm_seqs_destroy(dseqs_all); dseqs_all = nullptr; }
        // which slot needs what: the full set where sequences are scored (slot 0) or folds are trained, a shard
        // where the main EM run is sharded
        for (size_t d = 0; d < ndev; d++) {
            Dev& dv = devs[d];
            const bool want_full = (in_em_group(d) && !sharded) || (d == 0 && o.score) || runs_folds(d);
            if (want_full) {
                if (d == 0 && dseqs_all) dv.full = dseqs_all;   // nothing was dropped: the seeding copy is the training set
                else if (bamm_seqs_upload(dv.ctx, use, 0, use->n_seqs, &dv.full)) die_abi("upload");
            } else if (d == 0 && dseqs_all) {
                bamm_seqs_destroy(dseqs_all);
            }
            if (sharded && in_em_group(d)) {
                if (bamm_shard_range(use->len, use->n_seqs, seeds.max_w, (uint32_t)d, (uint32_t)ne, &dv.begin, &dv.end)) die_abi("shard range");
                if (bamm_seqs_upload(dv.ctx, use, dv.begin, dv.end, &dv.shard)) die_abi("upload of a shard");
            } else {
                dv.shard = dv.full; dv.begin = 0; dv.end = use->n_seqs;
            }
        }
        dseqs_all = nullptr;
        kept_len.assign(use->len, use->len + use->n_seqs);
        stage("device contexts + upload of the positives");
        if (sharded || o.forceComm) {
            const size_t nc = sharded ? ne : 1;
            std::vector<bamm_ctx*> ctxs;
            std::vector<bamm_comm*> comms(nc, nullptr);
            for (size_t d = 0; d < nc; d++) ctxs.push_back(devs[d].ctx);
            if (distinct) {
                if (bamm_comm_init_all(ctxs.data(), (uint32_t)nc, comms.data())) die_abi("RCCL communicator");
            } else {                                         // the largest buffer summed: the count table + 3, or EM::mask's histogram
                const uint64_t words = std::max<uint64_t>((uint64_t)seeds.max_w * (uint64_t(1) << (2 * (o.K + 1))) + 3, 2049);
                if (bamm_comm_init_local(ctxs.data(), (uint32_t)nc, words, comms.data())) die_abi("host-staged communicator");
            }
            for (size_t d = 0; d < nc; d++) devs[d].comm = comms[d];
            stage(distinct ? "RCCL communicator over the GPUs" : "host-staged communicator over the contexts");
        }
        if (o.score || o.FDR) {
            // negative set sampled from the s-mer statistics of the (kept) positives, mainBaMM.cpp:97-116.  The sampler
            // (host, all cores), the packing and the upload run on a thread of their own BESIDE the seeding and the main
            // EM run, which need none of it; the first consumer -- the folds, --scoreSeqset -- waits (ensure_negatives).
            size_t mFold = o.mFold;
            const size_t minSeqN = 5000;
            if (posN < minSeqN) mFold = minSeqN / posN + (minSeqN % posN ? 1 : 0);
            const size_t n_pos = use->n_seqs;
            negN = n_pos * mFold;
            // the folds of --FDR score every cvFold-th negative and nothing else (FDR.cpp:58-60): without --scoreSeqset
            // only those are generated, packed and uploaded (the others still consume their draws of the stream)
            const size_t stride = (o.FDR && !o.score) ? cvF : 0;
            neg_thread = std::thread([&, use, filtered, n_pos, mFold, stride] {
                auto fail_abi = [&](const char* what) { neg_err = std::string("Error: ") + what + ": " + bamm_last_error(); };
                auto t0 = std::chrono::high_resolution_clock::now();
                std::string serr;
                bamm_packed* npk = nullptr;
                // the sampler on the device (csrc/negs.hip) where the kept positives are resident on slot 0 and the
                // negatives are wanted as a set of their own -- all of them, or the folds' subset; it declines (-s other than 2,
                // a libc that is not glibc, ...) with BAMM_ERR_UNSUPPORTED and the host path below takes over
                // (--scoreSeqset --saveLogOdds prints the negatives' text: the host path keeps their codes)
                size_t dfull = ndev;                             // the first slot that holds every kept positive (with a sharded main
                for (size_t d = 0; d < ndev && dfull == ndev; d++) if (devs[d].full) dfull = d;   // run: a fold's slot)
                if (!o.hostSampler && dfull < ndev && (stride > 1 || !o.FDR) && !(o.score && o.saveLogOdds)) {
                    const int rc = bamm_sample_negatives(devs[dfull].ctx, devs[dfull].full, (uint32_t)o.sOrder, mFold, o.genericNeg ? 1 : 0, stride, &npk, nullptr);
                    if (rc != BAMM_OK && rc != BAMM_ERR_UNSUPPORTED) return fail_abi("negative sampler");
                    if (rc == BAMM_OK) {
                        neg_on_device = true;
                        if (filtered) bamm_packed_free(filtered);
                        neg_off.assign(1, 0);
                        for (uint64_t n = 0; n < npk->n_seqs; n++) neg_off.push_back(neg_off.back() + npk->len[n]);
                    }
                }
                if (!npk)
                {
                    std::vector<uint32_t, DefaultInitAlloc<uint32_t>> ys(use->total_len ? use->total_len : 1);   // every cell is written
                    std::vector<uint64_t> uoff(n_pos + 1, 0);
                    int rc = bamm_unpack_y(use, (uint32_t)o.sOrder, ys.data());
                    for (uint64_t n = 0; n < n_pos; n++) uoff[n + 1] = uoff[n] + use->len[n];
                    if (filtered) bamm_packed_free(filtered);     // the thread is the last reader of the kept positives' packing
                    if (rc) return fail_abi("unpack");
                    if (sample_negatives(ys.data(), uoff.data(), n_pos, (uint32_t)o.sOrder, mFold, o.genericNeg, neg_codes, neg_off, serr, stride)) { neg_err = serr; return; }
                }
                auto t1 = std::chrono::high_resolution_clock::now();
                neg_t_sample = std::chrono::duration<double>(t1 - t0).count();
                if (!npk && bamm_pack_codes(neg_codes.data(), neg_off.data(), neg_off.size() - 1, 1, &npk)) return fail_abi("packing negatives");
                if (stride > 1) {                                // what was sampled IS the folds' subset
                    for (size_t n = 0; n + 1 < neg_off.size(); n++) neg_cv_len.push_back((uint32_t)(neg_off[n + 1] - neg_off[n]));
                    for (size_t d = 0; d < ndev; d++)
                        if (runs_folds(d) && bamm_seqs_upload(devs[d].ctx, npk, 0, npk->n_seqs, &devs[d].neg_cv)) return fail_abi("upload negatives");
                } else {
                    if (bamm_seqs_upload(devs[0].ctx, npk, 0, npk->n_seqs, &devs[0].neg)) return fail_abi("upload negatives");
                    if (o.FDR) {                                 // the folds' subset as a set of its own, on the slots that run folds
                        std::vector<uint64_t> sub_off{0};
                        ByteVec sub_codes;
                        for (size_t i = 0; i + cvF <= negN; i += cvF) {
                            sub_codes.insert(sub_codes.end(), neg_codes.begin() + (ptrdiff_t)neg_off[i], neg_codes.begin() + (ptrdiff_t)neg_off[i + 1]);
                            sub_off.push_back(sub_codes.size());
                            neg_cv_len.push_back((uint32_t)(neg_off[i + 1] - neg_off[i]));
                        }
                        bamm_packed* spk = nullptr;
                        if (bamm_pack_codes(sub_codes.data(), sub_off.data(), sub_off.size() - 1, 1, &spk)) return fail_abi("packing negatives");
                        for (size_t d = 0; d < ndev; d++)
                            if (runs_folds(d) && bamm_seqs_upload(devs[d].ctx, spk, 0, spk->n_seqs, &devs[d].neg_cv)) return fail_abi("upload negatives");
                        bamm_packed_free(spk);
                    }
                }
                bamm_packed_free(npk);
                neg_t_pack = std::chrono::duration<double>(std::chrono::high_resolution_clock::now() - t1).count();
            });
        } else if (filtered) {
            bamm_packed_free(filtered);
        }
    }
    // scorer over a resident set: MOPS scores (concatenated), ZOOPS maxima
    auto score_set = [&](bamm_ctx* ctx, bamm_seqs* set, const std::vector<uint32_t>& lens, const Motif& m, std::vector<float>& mops,
                         std::vector<float>& zoops, const uint8_t* subset = nullptr, bool want_mops = true,
                         std::vector<uint64_t>* z_out = nullptr) {
        size_t total = 0;
        if (want_mops) for (uint32_t L : lens) total += L - m.W + 1;
        mops.assign(total ? total : 1, 0.f);
        zoops.assign(lens.size() ? lens.size() : 1, 0.f);
        std::vector<uint64_t> z_local;
        std::vector<uint64_t>& z = z_out ? *z_out : z_local;
        z.assign(lens.size() ? lens.size() : 1, 0);
        if (bamm_logodds_subset(ctx, set, subset, m.K, m.W, bg.K, m.v.data(), bg.v.data(), want_mops ? mops.data() : nullptr, total,
                                zoops.data(), z.data())) return 1;
        mops.resize(total);
        zoops.resize(lens.size());
        return 0;
    };
    // main thread only, before the first consumer of the negative set (the folds, --scoreSeqset)
    auto ensure_negatives = [&]() {
        if (!neg_thread.joinable()) return;
        neg_thread.join();
        if (!neg_err.empty()) die(neg_err);
        if (o.score) for (size_t n = 0; n + 1 < neg_off.size(); n++) neg_len.push_back((uint32_t)(neg_off[n + 1] - neg_off[n]));
        if (timing) std::cerr << "[timing-beside] negative set: sample (" << (neg_on_device ? "device" : "host") << ", rand() stream of the reference) " << neg_t_sample
                              << " s, pack + upload " << neg_t_pack << " s, on a thread of their own beside the stages above" << std::endl;
        stage("negative set: wait for the sampler thread");
    };
    // kept positives: FASTA codes / headers in the same order as the resident set -- what --scoreSeqset's writers print;
    // copies only where a record was dropped (0.1 s at a million records otherwise, for nothing)
    std::vector<std::string> kept_headers_own;
    ByteVec kept_codes_own;
    std::vector<uint64_t> kept_off_own{0};
    if (o.score && posN != pos.size())
        for (size_t n = 0; n < pos.size(); n++)
            if (keep[n]) {
                kept_headers_own.push_back(pos.headers[n]);
                kept_codes_own.insert(kept_codes_own.end(), pos.codes.begin() + pos.off[n], pos.codes.begin() + pos.off[n + 1]);
                kept_off_own.push_back(kept_codes_own.size());
            }
    const bool kept_all = posN == pos.size();
    const std::vector<std::string>& kept_headers = kept_all ? pos.headers : kept_headers_own;
    const ByteVec& kept_codes = kept_all ? pos.codes : kept_codes_own;
    const std::vector<uint64_t>& kept_off = kept_all ? pos.off : kept_off_own;
    auto em_params = [&](const Motif& m) {
        bamm_em_params p;
        bamm_em_default_params(&p);
        p.K = m.K; p.W = m.W; p.bg_order = bg.K; p.q = m.q; p.optimize_q = o.optimizeQ;
        p.epsilon = o.epsilon; p.max_iterations = o.max_iter;
        p.n_seqs_bound = posN;                               // one unit for the count accumulator on every GPU
        return p;
    };

    // ---- --FDR: the folds of one motif (FDR.cpp:37-127).  Fold f trains on the positives {n : n mod cvFold != f} from the SEED
    // model and scores its test positives and every cvFold-th negative on slot fold_slot[f]; one host thread per slot in
    // use (the reference runs its folds on OpenMP threads, FDR.cpp:37); every fold keeps its scores to itself and they
    // are merged in fold order afterwards, so the files do not depend on the plan or on which fold finishes first.
    struct FoldOut { std::vector<float> posMax, negMax, posAll, negAll; float q = 0.f; std::string log, err; };
    std::vector<std::vector<FoldOut>> fold_results(seeds.motifs.size());
    auto run_folds = [&](size_t n, std::vector<FoldOut>& folds) {
        const size_t cv = o.cvFold, P = kept_len.size();
        const Motif& seed = seeds.motifs[n];
        folds.assign(cv, FoldOut());
        for (auto& f : folds) f.q = seed.q;
        std::vector<size_t> slots_in_use;
        for (size_t f = 0; f < cv; f++)
            if (std::find(slots_in_use.begin(), slots_in_use.end(), fold_slot[f]) == slots_in_use.end()) slots_in_use.push_back(fold_slot[f]);
        auto one_fold = [&](size_t fold) {
            Dev& dv = devs[fold_slot[fold]];
            FoldOut& fo = folds[fold];
            Motif m = seed;
            std::vector<uint8_t> train(P, 0), test(P, 0);
            for (size_t i = 0; i + cv <= P; i += cv)         // strided split; the last P mod cv records are unused
                for (size_t f = 0; f < cv; f++) (f != fold ? train : test)[i + f] = 1;
            if (o.EM) {
                bamm_em_params p = em_params(m);
                bamm_em* em = nullptr;
                if (bamm_em_create(dv.ctx, dv.full, &p, bg.v.data(), m.A.data(), m.v.data(), train.data(), &em)) { fo.err = bamm_last_error(); return; }
                uint32_t it = 0;
                auto t0 = std::chrono::high_resolution_clock::now();
                int rc;
                if (!o.advanceEM) rc = bamm_em_optimize(em, &it);                              // FDR.cpp:67-72
                else rc = bamm_em_mask(em, o.f, &it, nullptr, nullptr);
                if (rc) { fo.err = bamm_last_error(); bamm_em_destroy(em); return; }
                bamm_em_get_v(em, m.v.data());
                bamm_em_get_q(em, &fo.q);
                bamm_em_destroy(em);
                std::ostringstream os;
                os << "\n--- Runtime for EM: " << std::chrono::duration<double>(std::chrono::high_resolution_clock::now() - t0).count() << " seconds ---\n";
                fo.log = os.str();
            }
            std::vector<float> mops, zoops;
            if (score_set(dv.ctx, dv.full, kept_len, m, mops, zoops, test.data(), o.mops)) { fo.err = bamm_last_error(); return; }
            size_t o_m = 0;
            for (size_t i = 0; i < P; i++) {
                const size_t nw = kept_len[i] - m.W + 1;
                if (test[i]) {
                    if (o.mops) fo.posAll.insert(fo.posAll.end(), mops.begin() + o_m, mops.begin() + o_m + nw);
                    if (o.zoops) fo.posMax.push_back(zoops[i]);
                }
                o_m += nw;
            }
            // negSet = every cv-th negative (FDR.cpp:58-60): resident as a set of its own, scored as a whole
            if (score_set(dv.ctx, dv.neg_cv, neg_cv_len, m, mops, zoops, nullptr, o.mops)) { fo.err = bamm_last_error(); return; }
            if (o.mops) fo.negAll = mops;
            if (o.zoops) fo.negMax = zoops;
        };
        std::vector<std::thread> team;
        for (size_t slot : slots_in_use)
            team.emplace_back([&, slot] { for (size_t f = 0; f < cv; f++) if (fold_slot[f] == slot) one_fold(f); });
        for (auto& t : team) t.join();
    };

    for (size_t n = 0; n < seeds.motifs.size(); n++) {
        Motif motif = seeds.motifs[n];                       // deep copy (mainBaMM.cpp:121)
        const std::string mbase = o.basename + "_motif_" + std::to_string(n + 1);
        if (o.saveInitial && motif_write(o.out_dir, o.basename + "_init_motif_" + std::to_string(n + 1), motif, err)) die(err);
        std::thread& fold_thread = g_fold_thread;            // overlap mode: this motif's folds train while its main run does (joined by die() too)
        if (overlap) ensure_negatives();                     // the folds score negatives
        if (overlap) fold_thread = std::thread([&, n] { run_folds(n, fold_results[n]); });
        struct FoldJoin { std::thread& t; ~FoldJoin() { if (t.joinable()) t.join(); } } fold_join{fold_thread};
        if (o.EM) {
            auto t0 = std::chrono::high_resolution_clock::now();
            const bamm_em_params p = em_params(motif);
            // one handle per GPU over its shard; with several GPUs each is driven by a host thread of its own and
            // every pass ends in one RCCL all-reduce, after which all of them hold the same model
            std::vector<bamm_em*> ems(ndev, nullptr);
            std::vector<std::string> thread_err(ndev);
            std::vector<uint32_t> its(ndev, 0);
            for (size_t d = 0; d < ne; d++) {
                if (d > 0 && !sharded) break;
                if (bamm_em_create(devs[d].ctx, devs[d].shard, &p, bg.v.data(), motif.A.data(), motif.v.data(), nullptr, &ems[d])) die_abi("EM");
                if (devs[d].comm && bamm_em_set_comm(ems[d], devs[d].comm)) die_abi("EM communicator");
            }
            const auto t_created = std::chrono::high_resolution_clock::now();
            // one std::thread per rank, not an OpenMP team (which may come back smaller than asked for and leave ranks
            // out of the collective); a rank that still fails aborts every communicator so that its peers return
            auto run_rank = [&](size_t d) {
                int rc;
                if (!o.advanceEM) rc = bamm_em_optimize(ems[d], &its[d]);                    // mainBaMM.cpp:133-137
                else rc = bamm_em_mask(ems[d], o.f, &its[d], nullptr, nullptr);
                if (rc) {
                    thread_err[d] = bamm_last_error();                                        // thread-local message
                    for (auto& dv : devs) if (dv.comm) bamm_comm_abort(dv.comm);
                }
            };
            if (sharded) {
                std::vector<std::thread> team;
                for (size_t d = 0; d < ne; d++) if (ems[d]) team.emplace_back(run_rank, d);
                for (auto& t : team) t.join();
            } else if (ems[0]) {
                run_rank(0);
            }
            for (size_t d = 0; d < ndev; d++)
                if (!thread_err[d].empty()) die("Error: EM on GPU " + std::to_string(devs[d].device) + ": " + thread_err[d]);
            bamm_em* em = ems[0];
            const uint32_t it = its[0];
            const auto t_optimized = std::chrono::high_resolution_clock::now();
            if (bamm_em_get_v(em, motif.v.data())) die_abi("get_v");
            float q = 0;
            bamm_em_get_q(em, &q);
            motif.q = q;
            if (o.verbose) {                                 // the lines EM.cpp:112-115 prints
                std::vector<float> llh(it), vd(it), qq(it);
                uint32_t cnt = 0;
                bamm_em_get_trace(em, llh.data(), vd.data(), qq.data(), it, &cnt);
                for (uint32_t i = 0; i < cnt && i < it; i++) {
                    if (o.advanceEM) {                        // EM.cpp:487
                        std::cout << i + 1 << "th iteration, delta_llikelihood=" << llh[i] - (i ? llh[i - 1] : 0.f) << std::endl;
                        continue;
                    }
                    if (o.optimizeQ && i < 5) std::cout << "optimized q=" << qq[i] << std::endl;
                    std::cout << i + 1 << " iter, llh=" << llh[i] << ", diff_llh=" << llh[i] - (i ? llh[i - 1] : 0.f)
                              << ", v_diff=" << vd[i] << std::endl;
                }
            }
            motif_calculate_p(motif, bg);
            auto dt = std::chrono::duration<double>(std::chrono::high_resolution_clock::now() - t0);
            std::cout << "\n--- Runtime for EM: " << dt.count() << " seconds ---\n";        // EM.cpp:134
            if (timing) std::cerr << "[timing-beside] EM of motif " << n + 1 << ": create " << std::chrono::duration<double>(t_created - t0).count()
                                  << " s, " << (o.advanceEM ? "mask" : "optimize") << " " << std::chrono::duration<double>(t_optimized - t_created).count()
                                  << " s (" << it << " passes), read-back + calculateP " << (dt - std::chrono::duration<double>(t_optimized - t0)).count() << " s" << std::endl;
            stage("EM (create + optimize + read-back)");
            if (o.saveBaMMs) {                                // EM::write (EM.cpp:553-601)
                std::vector<float> cnts(bamm_v_size(motif.K, motif.W));
                bamm_em_get_counts(em, cnts.data());
                std::ofstream fn(o.out_dir + '/' + mbase + ".counts");
                for (uint32_t j = 0; j < motif.W; j++) {
                    for (uint32_t k = 0; k <= motif.K; k++) {
                        for (size_t y = 0; y < (size_t(1) << (2 * (k + 1))); y++)
                            fn << static_cast<int>(cnts[bamm_v_offset(k, motif.W) + y * motif.W + j]) << '\t';
                        fn << std::endl;
                    }
                    fn << std::endl;
                }
                // r of every kept sequence, shard after shard (the shards are consecutive ranges)
                uint64_t total = 0;
                for (uint32_t L : kept_len) total += L;
                std::vector<float> r(total ? total : 1);
                uint64_t ro_base = 0;
                for (size_t d = 0; d < ndev; d++) {
                    if (!ems[d]) continue;
                    uint64_t ns = 0, tl = 0;
                    bamm_seqs_info(devs[d].shard, &ns, &tl, nullptr, nullptr);
                    if (tl && bamm_em_get_r(ems[d], 0, ns, r.data() + ro_base, tl)) die_abi("getR");
                    ro_base += tl;
                }
                std::ofstream fp(o.out_dir + '/' + mbase + ".positions");
                fp << "seq\tlength\tstrand\tstart..end\tpattern" << std::endl;
                static const char B[] = "NACGT";
                uint64_t ro = 0;
                for (size_t s = 0; s < pos.size(); s++) {
                    if (!keep[s]) continue;
                    const size_t Lfull = packed->len[s], L0 = pos.off[s + 1] - pos.off[s];
                    const size_t Lshown = o.ss ? Lfull : (Lfull - 1) / 2;
                    for (size_t i = 0; i + motif.W <= Lfull; i++) {
                        if (r[ro + Lfull - motif.W - i] >= 0.3f) {
                            fp << pos.headers[s] << '\t' << Lshown << '\t' << ((i < Lshown) ? '+' : '-') << '\t' << i + 1 << ".." << i + motif.W << '\t';
                            for (size_t b = i; b < i + motif.W; b++) {
                                uint8_t code;                 // Sequence::getSequence(): forward, N, reverse complement
                                if (b < L0) code = pos.codes[pos.off[s] + b];
                                else if (o.ss || b == L0) code = 0;
                                else { const uint8_t c = pos.codes[pos.off[s] + (2 * L0 - b)]; code = (c >= 1 && c <= 4) ? (uint8_t)(5 - c) : 0; }
                                fp << B[code];
                            }
                            fp << std::endl;
                        }
                    }
                    ro += Lfull;
                }
            }
            std::cout << "optimized q = " << q << std::endl;   // mainBaMM.cpp:147
            for (bamm_em* e : ems) bamm_em_destroy(e);
        } else {
            std::cout << "Note: the model is not optimized!\n";
        }
        if (motif_write(o.out_dir, mbase, motif, err)) die(err);
        stage("write model (+ .counts/.positions)");
        if (o.score) ensure_negatives();
        if (o.score) {                                       // mainBaMM.cpp:171-236
            if (o.verbose) std::cout << std::endl << "*************************" << std::endl << "*    Score Sequences    *" << std::endl << "*************************" << std::endl << std::endl;
            Motif sm = motif;
            if (!o.EM && o.seed_tag == "BaMM" && o.bg_file.empty()) die("No background Model file provided for initial search motif!");
            std::vector<float> neg_mops, neg_zoops, pos_mops, pos_zoops, pv, ev;
            std::vector<uint64_t> neg_z, pos_z;
            if (score_set(devs[0].ctx, devs[0].neg, neg_len, sm, neg_mops, neg_zoops, nullptr, true, &neg_z)) die_abi("calcLogOdds");
            if (score_set(devs[0].ctx, devs[0].full, kept_len, sm, pos_mops, pos_zoops, nullptr, true, &pos_z)) die_abi("calcLogOdds");
            if (o.saveLogOdds) {                             // mainBaMM.cpp:204-208, :223-227
                const std::vector<std::string> neg_headers(negN, "> bg_seq");                     // SeqGenerator.cpp:228
                if (logodds_zoops_write(o.out_dir, o.basename + ".negSet", neg_headers, neg_codes.data(), neg_off.data(), negN, false,
                                        o.ss, sm.W, neg_zoops.data(), neg_z.data(), err)) die(err);
                if (logodds_zoops_write(o.out_dir, mbase, kept_headers, kept_codes.data(), kept_off.data(), kept_len.size(), !o.ss,
                                        o.ss, sm.W, pos_zoops.data(), pos_z.data(), err)) die(err);
            }
            mops_pvalues(pos_mops.data(), pos_mops.size(), neg_mops, kept_len.size(), pv, ev);
            if (occurrence_write(o.out_dir, mbase, kept_headers, kept_codes.data(), kept_off.data(), kept_len.size(), o.ss, sm.W,
                                 pv.data(), ev.data(), o.pvalCutoff, err)) die(err);
            stage("--scoreSeqset: score + p-values + .occurrence");
        }
    }

    ensure_negatives();
    if (o.FDR) {                                             // mainBaMM.cpp:243-265, FDR.cpp:28-145
        if (o.verbose) std::cout << std::endl << "***********************" << std::endl << "*   BaMM validation   *" << std::endl << "***********************" << std::endl;
        const size_t cv = o.cvFold, P = kept_len.size();
        for (size_t n = 0; n < seeds.motifs.size(); n++) {
            const Motif& seed = seeds.motifs[n];
            // the folds of this motif (run_folds above): trained while the main run was training (overlap mode), else here
            if (fold_results[n].empty()) run_folds(n, fold_results[n]);
            std::vector<FoldOut>& folds = fold_results[n];
            std::vector<float> posMax, negMax, posAll, negAll;
            float updatedQ = seed.q;
            for (size_t fold = 0; fold < cv; fold++) {        // merge in fold order
                FoldOut& fo = folds[fold];
                if (!fo.err.empty()) die("Error: fold " + std::to_string(fold) + ": " + fo.err);
                std::cout << fo.log;
                posMax.insert(posMax.end(), fo.posMax.begin(), fo.posMax.end());
                negMax.insert(negMax.end(), fo.negMax.begin(), fo.negMax.end());
                posAll.insert(posAll.end(), fo.posAll.begin(), fo.posAll.end());
                negAll.insert(negAll.end(), fo.negAll.begin(), fo.negAll.end());
                if (o.EM) updatedQ = fo.q;                    // the reference keeps whichever fold wrote last (FDR.cpp:73): the last one here
            }
            stage("--FDR: fold EMs + scoring (GPU)");
            const std::string fbase = o.basename + "_motif_" + std::to_string(n + 1);
            if (o.saveLogOdds && fdr_logodds_write(o.out_dir, fbase, posMax, negMax, posAll, negAll, P, negN, o.mops, o.zoops,
                                                   o.savePvalues, err)) die(err);
            FdrResult res;
            fdr_statistics(posMax, negMax, posAll, negAll, P, negN, updatedQ, o.mops, o.zoops, o.savePvalues, res);
            if (fdr_write(o.out_dir, fbase, res, P, negN, o.mops, o.zoops, o.savePRs, o.savePvalues, err)) die(err);
            stage("--FDR: PR / p-value statistics + writers (host)");
        }
    }

    std::cout << std::endl << "******************" << std::endl << "*   Statistics   *" << std::endl << "******************" << std::endl;
    std::cout << "Alphabet type is ACGT";                     // Global::printStat (Global.cpp:346-392)
    std::cout << "\nGiven initial model is " << base_name(o.seed_file) << ", BaMM order: " << o.K << ", bgmodel order: " << o.Kbg;
    std::cout << "\nBaMM is learned from " << (o.ss ? "single-stranded sequences." : "double-stranded sequences.");
    std::cout << "\nGiven positive sequence set is " << o.basename << ".\n	" << pos.size() << " sequences, max.length: " << pos.max_len
              << ", min.length: " << pos.min_len << "\n	base frequencies:";
    for (int i = 0; i < 4; i++) std::cout << ' ' << pos.base_freq[i] << "(" << "ACGT"[i] << ")";
    if (o.advanceEM) std::cout << "\n    " << o.f * 100 << "% of the sequences are used for EM after masking.";   // Global.cpp:370-372
    std::cout << "\nThe background model is generated based on cond.prob of " << o.sOrder << "-mers.";
    if (o.FDR) std::cout << "\nFolds for cross-validation (FDR estimation): " << o.cvFold;
    auto dt = std::chrono::duration<double>(std::chrono::high_resolution_clock::now() - t0_wall);
    std::cout << std::endl << "------ Runtime: " << dt.count() << " seconds -------" << std::endl;

    // Everything is written and closed.  What is left is giving memory back -- a dozen hipFree calls (each a device
    // synchronisation), a hundred megabytes of host vectors, then the HIP runtime's own static destructors: 0.1 s of a
    // 0.7 s command that ends anyway.  The process leaves here (no other thread is alive: the side threads were joined
    // where their results were taken); --debug keeps the orderly teardown for leak checkers.
    if (timing) fprintf(stderr, "[timing-abs] main left at %.4f\n", epoch());
    if (!o.debug) {
        // (every writer of this file is a scoped std::ofstream / FILE closed where its stage ends; tests/test_cli_gpu.py compares
        // the files of a --debug run, which takes the orderly way out below, byte for byte with this one's)
        for (auto& dv : devs)
            if (dv.comm) { bamm_comm_destroy(dv.comm); dv.comm = nullptr; }      // peers of a sharded run are told, not left waiting
        std::cout.flush(); std::cerr.flush();
        fflush(nullptr);
        _exit(0);
    }
    for (auto& dv : devs) {
        if (dv.comm) bamm_comm_destroy(dv.comm);
        if (dv.neg) bamm_seqs_destroy(dv.neg);
        if (dv.neg_cv) bamm_seqs_destroy(dv.neg_cv);
        if (dv.shard && dv.shard != dv.full) bamm_seqs_destroy(dv.shard);
        if (dv.full) bamm_seqs_destroy(dv.full);
        if (dv.ctx) bamm_ctx_destroy(dv.ctx);
    }
    bamm_packed_free(packed);
    if (timing) fprintf(stderr, "[timing-abs] teardown done at %.4f\n", epoch());
    return 0;
}
