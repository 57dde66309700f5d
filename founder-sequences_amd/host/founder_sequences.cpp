// founder_sequences -- command line front end over the C ABI (include/fseq.h).
//
// Keeps the option surface, defaults and validation messages of the reference CLI
// (founder-sequences/cmdline.ggo:12-29, founder-sequences/main.cc:68-150) and the controller's
// input checks and outputs (founder-sequences/generate_context.cc:64-106,161-200,393-433).
// Host C++17 only; all segmentation work happens behind fseq_run_segmentation on the GPU.
// Joining: greedy (greedy_matcher.cc), bipartite-matching (bipartite_matcher.cc; an own Kuhn-Munkres
// in place of Lemon 1.3.1's MaxWeightedPerfectMatching) and random (join_context.cc:259-289).
#include <fseq.h>
#include "fseq_shard_rccl.hpp"

#include <getopt.h>

#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <iostream>
#include <limits>
#include <string>
#include <vector>

namespace {

enum class joining { BIPARTITE_MATCHING, GREEDY, RANDOM };
enum class input_format { FASTA, LIST_FILE };

char const *const USAGE =
	"Usage: founder_sequences --input=input-list.txt --segment-length-bound=... --output-founders=...\n"
	"Generate a segmentation in O(mn log sigma) time and output founder sequences.\n\n"
	"  -h, --help                         Print help and exit\n"
	"  -V, --version                      Print version and exit\n"
	"\nInput and output options:\n"
	"  -i, --input=PATH                   Input file path\n"
	"  -f, --input-format=FORMAT          Input file format  (possible values=\"FASTA\", \"list-file\" default=`list-file')\n"
	"  -e, --output-segments=PATH         Output segment co-ordinates in text format\n"
	"  -o, --output-founders=PATH         Founder file path\n"
	"\nAlgorithm parameters:\n"
	"  -s, --segment-length-bound=SIZE    Segment length bound\n"
	"  -j, --segment-joining=METHOD       Segment joining method  (possible values=\"bipartite-matching\", \"greedy\", \"random\" default=`bipartite-matching')\n"
	"\nRunning options:\n"
	"  -m, --pbwt-sample-rate=q           On the first pass, store a PBWT sample every q*sqrt(n)-th position. Zero indicates no sampling.  (default=`4')\n"
	"      --random-seed=LONG             Seed for the random number generator  (default=`0')\n"
	"      --single-threaded              Use only one worker thread  (default=off)\n"
	"      --print-invocation             Print the command line arguments to stderr  (default=off)\n"
	"      --gpus=N                       Shard the alignment over the first N GPUs of this node (RCCL)  (default=`1')\n";

bool read_file(std::string const &path, std::string &out)
{
	std::ifstream f(path, std::ios::binary);
	if (!f) return false;
	out.assign(std::istreambuf_iterator<char>(f), std::istreambuf_iterator<char>());
	return true;
}

// list-file: one path per line, each file holds one sequence (README.md:86)
bool read_list_file(char const *path, std::vector<std::string> &seqs)
{
	std::ifstream f(path);
	if (!f) { std::cerr << "Unable to open the input file '" << path << "'." << std::endl; return false; }
	std::string line;
	while (std::getline(f, line))
	{
		if (line.empty()) continue;
		std::string content;
		if (!read_file(line, content)) { std::cerr << "Unable to open the sequence file '" << line << "'." << std::endl; return false; }
		seqs.emplace_back(std::move(content));
	}
	return true;
}

// FASTA: '>' header lines, sequence lines concatenated (README.md:80)
bool read_fasta(char const *path, std::vector<std::string> &seqs)
{
	std::ifstream f(path);
	if (!f) { std::cerr << "Unable to open the input file '" << path << "'." << std::endl; return false; }
	std::string line;
	bool open = false;
	while (std::getline(f, line))
	{
		if (!line.empty() && line.back() == '\r') line.pop_back();
		if (!line.empty() && line[0] == '>') { seqs.emplace_back(); open = true; continue; }
		if (!open) { if (line.empty()) continue; seqs.emplace_back(); open = true; }
		seqs.back() += line;
	}
	return true;
}

std::ostream *open_out(char const *path, std::ofstream &file)
{
	if (!path || ('-' == path[0] && '\0' == path[1])) return &std::cout;
	file.open(path, std::ios::binary | std::ios::trunc);
	if (!file) { std::cerr << "Unable to open '" << path << "' for writing." << std::endl; std::exit(EXIT_FAILURE); }
	return &file;
}

} // namespace

int main(int argc, char **argv)
{
	char const *input = nullptr, *out_segments = nullptr, *out_founders = nullptr;
	input_format fmt = input_format::LIST_FILE;
	joining join = joining::BIPARTITE_MATCHING;
	long seg_len = 0, sample_rate = 4, seed = 0;
	bool seg_len_given = false, single_threaded = false, print_invocation = false;
	long gpus = 1;
	bool gpus_given = false;

	static option const longopts[] = {
		{"help", no_argument, nullptr, 'h'}, {"version", no_argument, nullptr, 'V'},
		{"input", required_argument, nullptr, 'i'}, {"input-format", required_argument, nullptr, 'f'},
		{"output-segments", required_argument, nullptr, 'e'}, {"output-founders", required_argument, nullptr, 'o'},
		{"segment-length-bound", required_argument, nullptr, 's'}, {"segment-joining", required_argument, nullptr, 'j'},
		{"pbwt-sample-rate", required_argument, nullptr, 'm'}, {"random-seed", required_argument, nullptr, 1000},
		{"single-threaded", no_argument, nullptr, 1001}, {"print-invocation", no_argument, nullptr, 1002},
		{"gpus", required_argument, nullptr, 1003},
		{nullptr, 0, nullptr, 0}};
	int c;
	while ((c = getopt_long(argc, argv, "hVi:f:e:o:s:j:m:", longopts, nullptr)) != -1)
	{
		switch (c)
		{
			case 'h': std::cout << USAGE; return EXIT_SUCCESS;
			case 'V': std::cout << "founder_sequences (MI355X build, fseq ABI " << fseq_abi_version() << ")\n"; return EXIT_SUCCESS;
			case 'i': input = optarg; break;
			case 'f':
				if (0 == strcmp(optarg, "FASTA")) fmt = input_format::FASTA;
				else if (0 == strcmp(optarg, "list-file")) fmt = input_format::LIST_FILE;
				else { std::cerr << argv[0] << ": invalid argument, \"" << optarg << "\", for option `--input-format' (`-f')" << std::endl; return EXIT_FAILURE; }
				break;
			case 'e': out_segments = optarg; break;
			case 'o': out_founders = optarg; break;
			case 's': seg_len = strtol(optarg, nullptr, 10); seg_len_given = true; break;
			case 'j':
				if (0 == strcmp(optarg, "bipartite-matching")) join = joining::BIPARTITE_MATCHING;
				else if (0 == strcmp(optarg, "greedy")) join = joining::GREEDY;
				else if (0 == strcmp(optarg, "random")) join = joining::RANDOM;
				else { std::cerr << argv[0] << ": invalid argument, \"" << optarg << "\", for option `--segment-joining' (`-j')" << std::endl; return EXIT_FAILURE; }
				break;
			case 'm': sample_rate = strtol(optarg, nullptr, 10); break;
			case 1000: seed = strtol(optarg, nullptr, 10); break;
			case 1001: single_threaded = true; break;
			case 1002: print_invocation = true; break;
			case 1003: gpus = strtol(optarg, nullptr, 10); gpus_given = true; break;
			default: return EXIT_FAILURE;
		}
	}
	if (!input) { std::cerr << argv[0] << ": '--input' ('-i') option required" << std::endl; return EXIT_FAILURE; }
	(void) single_threaded;

	// main.cc:80-115
	if (print_invocation)
	{
		std::cerr << "Invocation:";
		for (int i = 0; i < argc; ++i) std::cerr << ' ' << argv[i];
		std::cerr << std::endl;
	}
	if (seg_len_given)
	{
		if (seg_len <= 0) { std::cerr << "Segment length bound must be positive." << std::endl; return EXIT_FAILURE; }
	}
	else { std::cerr << "Segment length bound needs to be specified when generating a segmentation." << std::endl; return EXIT_FAILURE; }
	if (!(0 <= seed && (unsigned long) seed <= std::numeric_limits<std::uint_fast32_t>::max()))
	{ std::cerr << "Random seed out of bounds." << std::endl; return EXIT_FAILURE; }
	if (sample_rate <= 0) { std::cerr << "PBWT sample rate multiplier must be non-negative." << std::endl; return EXIT_FAILURE; }
	if (gpus < 1 || gpus > 64) { std::cerr << "The number of GPUs must be positive." << std::endl; return EXIT_FAILURE; }

	// generate_context.cc:64-106
	std::cerr << "Loading the input…" << std::flush;
	std::vector<std::string> seqs;
	if (!(input_format::FASTA == fmt ? read_fasta(input, seqs) : read_list_file(input, seqs))) return EXIT_FAILURE;
	if (seqs.empty()) { std::cerr << "\nThe input file contained no sequences." << std::endl; return EXIT_SUCCESS; }
	size_t const seq_length = seqs.front().size();
	std::cerr << " length: " << seq_length << std::endl;
	std::cerr << "Checking the input…" << std::endl;
	{
		bool stop = false;
		for (size_t i = 1; i < seqs.size(); ++i)
			if (seqs[i].size() != seq_length)
			{
				stop = true;
				std::cerr << "The length of the sequence at index " << i << " was " << seqs[i].size()
				          << " while that of the first one was " << seq_length << '.' << std::endl;
			}
		if (stop) return EXIT_FAILURE;
	}
	if (0 == seq_length) { std::cerr << "The sequences are empty." << std::endl; return EXIT_FAILURE; }

	fseq_params p{};
	p.m = (uint32_t) seqs.size();
	p.n = seq_length;
	p.segment_length = (uint64_t) seg_len;
	p.pbwt_sample_rate = (uint64_t) sample_rate;
	std::vector<uint8_t const *> rows(seqs.size());
	for (size_t i = 0; i < seqs.size(); ++i) rows[i] = reinterpret_cast<uint8_t const *>(seqs[i].data());
	bool const sharded = gpus > 1 && p.n >= 2 * p.segment_length;      // (the short path is one sweep: it does not shard)
	if (gpus > 1 && !sharded) std::cerr << "The sequences are shorter than two segments; using one GPU." << std::endl;

	// One context per rank.  --gpus N: every rank on its own thread and device, RCCL for the exchanges
	// (fseq_shard_rccl.hpp); the ranks make the same calls in the same order and get the same result.
	fseq_host::rccl_world world;
	std::vector<fseq_ctx *> ctxs(sharded ? (size_t) gpus : 1, nullptr);
	std::vector<fseq_result> results(ctxs.size());
	int rc = FSEQ_OK;
	if (sharded || gpus_given)
	{
		// (--gpus 1 still makes the communicator and runs the one-word all-reduce: the transport is checked wherever it is asked for)
		std::string err;
		if (!world.init(sharded ? (int) gpus : 1, err) || !world.self_test(err)) { std::cerr << err << std::endl; return EXIT_FAILURE; }
		std::cerr << "RCCL: " << world.world() << " rank(s), self-test passed." << std::endl;
	}
	std::cerr << "Generating a compressed alphabet…" << std::endl;
	std::cerr << "Calculating the segmentation…" << std::endl;
	auto run_rank = [&](int r) -> int {
		fseq_params pr(p);
		pr.device = r;
		int rc_(fseq_create(&pr, &ctxs[r]));
		if (FSEQ_OK == rc_ && sharded) rc_ = world.attach(ctxs[r], r);
		if (sharded)
		{
			// Before the first collective the rank threads agree on the host: a rank without a context or an exchange buffer
			// (no device memory, ...) cannot take part in a status exchange, so everybody leaves here with its code.  From
			// here on a failing rank posts its code in the exchange the others make next (fseq.h: FSEQ_E_PEER).
			int const worst(world.agree(rc_));
			if (FSEQ_OK != worst) return FSEQ_OK != rc_ ? rc_ : FSEQ_E_PEER;
		}
		else if (FSEQ_OK != rc_) return rc_;
		if (FSEQ_OK != (rc_ = fseq_set_rows(ctxs[r], rows.data()))) return rc_;     // (sharded: posts its own failures)
		return fseq_run_segmentation(ctxs[r], &results[r]);
	};
	if (sharded)
	{
		std::vector<int> const rcs(world.run(run_rank));
		for (size_t r = 0; r < rcs.size(); ++r)
			if (FSEQ_OK != rcs[r] && FSEQ_E_NO_REDUCTION != rcs[r])
			{
				// (a rank that failed on its own reports its error, the others FSEQ_E_PEER: print the cause)
				if (FSEQ_E_PEER != rcs[r] || FSEQ_OK == rc || FSEQ_E_PEER == rc) rc = rcs[r];
				std::cerr << "rank " << r << ": " << (ctxs[r] ? fseq_last_error(ctxs[r]) : fseq_strerror(rcs[r])) << std::endl;
			}
		if (FSEQ_OK == rc) rc = rcs[0];
		if (FSEQ_OK != rc && FSEQ_E_NO_REDUCTION != rc) return EXIT_FAILURE;
	}
	else
		rc = run_rank(0);
	fseq_ctx *ctx = ctxs[0];
	fseq_result res(results[0]);
	if (!ctx) { std::cerr << "Unable to initialise the GPU engine: " << fseq_strerror(rc) << std::endl; return EXIT_FAILURE; }
	if (FSEQ_E_NO_REDUCTION == rc)
	{
		// generate_context.cc:192-200
		std::cerr << "Unable to reduce the number of sequences; the maximum segment size is equal to the number of input sequences." << std::endl;
		return EXIT_FAILURE;
	}
	if (FSEQ_OK != rc) { std::cerr << fseq_last_error(ctx) << std::endl; return EXIT_FAILURE; }

	std::ofstream founders_file, segments_file;
	if (res.short_path)
	{
		// generate_context.cc:161-176, segmentation_sp_context.cc:31-47
		std::vector<uint32_t> first(res.max_segment_size), len(res.max_segment_size);
		fseq_short_path_runs(ctx, first.data(), len.data());
		std::cerr << "Outputting…" << std::endl;
		std::ostream &os = *open_out(out_founders, founders_file);
		for (uint32_t i = 0; i < res.max_segment_size; ++i) { os.write(seqs[first[i]].data(), (std::streamsize) seq_length); os << '\n'; }
		os << std::flush;
		if (out_segments)
		{
			std::ostream &ss = *open_out(out_segments, segments_file);
			ss << "SEQUENCE" "\n";
			for (uint32_t i = 0; i < res.max_segment_size; ++i) ss << len[i] << '\n';
			ss << std::flush;
		}
		std::cerr << "Done." << std::endl;
		fseq_destroy(ctx);
		return EXIT_SUCCESS;
	}

	// generate_context.cc:224-228
	std::cerr << "After calculating the traceback there were " << res.dp_segment_count
	          << " segments the maximum size of which was " << res.max_segment_size << '.' << std::endl;
	std::cerr << "Joining the remaining segments…" << std::endl;
	std::vector<uint32_t> perm((size_t) res.segment_count * res.max_segment_size);
	std::vector<uint32_t> all_a, all_d;                         // sharded: the boundary states, collected from their owners
	std::vector<fseq_segment> segments;
	if (sharded)
	{
		size_t const S(res.segment_count), m(p.m);
		segments.resize(S);
		if (FSEQ_OK != (rc = fseq_get_segments(ctx, segments.data()))) { std::cerr << fseq_last_error(ctx) << std::endl; return EXIT_FAILURE; }
		all_a.resize(S * m); all_d.resize(S * m);
		std::vector<uint64_t> lbs(S), rbs(S);
		for (size_t i = 0; i < S; ++i)
		{
			lbs[i] = segments[i].lb; rbs[i] = segments[i].rb;
			uint32_t owner(0);
			if (FSEQ_OK != (rc = fseq_shard_owner(ctx, segments[i].rb, &owner)) ||
			    FSEQ_OK != (rc = fseq_boundary_state(ctxs[owner], i, all_a.data() + i * m, all_d.data() + i * m)))
			{ std::cerr << fseq_last_error(ctxs[owner < ctxs.size() ? owner : 0]) << std::endl; return EXIT_FAILURE; }
		}
		switch (join)
		{
			case joining::GREEDY: rc = fseq_greedy_match_host(p.m, res.max_segment_size, S, lbs.data(), rbs.data(), all_a.data(), all_d.data(), perm.data()); break;
			case joining::BIPARTITE_MATCHING: rc = fseq_bipartite_match_host(p.m, res.max_segment_size, S, lbs.data(), rbs.data(), all_a.data(), all_d.data(), perm.data(), nullptr); break;
			case joining::RANDOM: rc = fseq_random_join_host(p.m, res.max_segment_size, S, lbs.data(), rbs.data(), all_a.data(), all_d.data(), (uint32_t) seed, perm.data()); break;
		}
	}
	else
	switch (join)                                               // join_context.cc:130-160
	{
		case joining::GREEDY: rc = fseq_join_greedy(ctx, perm.data()); break;
		case joining::BIPARTITE_MATCHING: rc = fseq_join_bipartite(ctx, perm.data()); break;
		case joining::RANDOM: rc = fseq_join_random(ctx, (uint32_t) seed, perm.data()); break;
	}
	if (FSEQ_OK != rc) { std::cerr << fseq_last_error(ctx) << std::endl; return EXIT_FAILURE; }
	std::cerr << "Outputting the founders…" << std::endl;
	// (not sharded: the lines are put together where the alignment already is, on the device; sharded: a rank holds its own columns)
	rc = sharded ? fseq_write_founders(ctx, rows.data(), perm.data(), out_founders) : fseq_write_founders_device(ctx, perm.data(), out_founders);
	if (FSEQ_OK != rc) { std::cerr << fseq_last_error(ctx) << std::endl; return EXIT_FAILURE; }
	if (out_segments)
	{
		// segmentation_dp_arg.cc:13-104; with greedy joining the copy-number matrix is empty, so only the
		// header is written (join_context.cc:57-61, greedy_matcher.cc:468-476)
		std::cerr << "Outputting the segments…" << std::endl;
		int const how = joining::GREEDY == join ? FSEQ_JOIN_GREEDY : (joining::RANDOM == join ? FSEQ_JOIN_RANDOM : FSEQ_JOIN_BIPARTITE);
		rc = sharded ? fseq_write_segments_host(ctx, rows.data(), how, all_a.data(), all_d.data(), out_segments)
		             : fseq_write_segments(ctx, rows.data(), how, out_segments);
		if (FSEQ_OK != rc) { std::cerr << fseq_last_error(ctx) << std::endl; return EXIT_FAILURE; }
	}
	std::cerr << "Done." << std::endl;
	for (fseq_ctx *c_ : ctxs) if (c_) fseq_destroy(c_);
	return EXIT_SUCCESS;
}
