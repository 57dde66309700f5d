// match_founder_sequences -- same command line surface and output as the reference's tool
// (match-sequences-to-founders/cmdline.ggo:4-16, match_founder_sequences.cc:108-259): thread every input
// sequence through the founders greedily (keep the set of founders that still match; when it runs empty, or
// -- with --min-segment-length -- as soon as the current piece has that length, print the piece and start
// again from all founders) and print SEQUENCE_INDEX LB RB FOUNDER_INDICES.  One task per sequence on the host's
// threads (one thread with --single-threaded), reports in input order.  Host only.
#include "aux_common.hpp"

#include <algorithm>
#include <atomic>
#include <cstring>
#include <getopt.h>
#include <mutex>
#include <numeric>
#include <sstream>
#include <thread>

namespace {

bool read_founders(char const *path, int format, std::vector<std::string> &out)
{
	if (2 == format)                                                            // list-file
	{
		std::ifstream f(path);
		if (!f) return false;
		std::vector<std::string> names;
		aux::read_lines(f, names);
		for (auto const &n : names) { out.emplace_back(); if (!aux::read_file(n, out.back())) return false; }
		return true;
	}
	std::ifstream f(path);
	if (!f) return false;
	std::string line;
	if (1 == format) { while (std::getline(f, line)) out.push_back(line); return true; }   // text: one sequence per line
	bool open = false;                                                          // FASTA
	while (std::getline(f, line))
	{
		if (!line.empty() && '>' == line[0]) { out.emplace_back(); open = true; }
		else if (open) out.back() += line;
	}
	return true;
}

} // namespace

int main(int argc, char **argv)
{
	char const *sequences = nullptr, *founders_path = nullptr;
	int format = 2;
	long min_len = 0;
	bool single_threaded = false;
	static option const opts[] = {{"sequences", required_argument, nullptr, 's'}, {"founders", required_argument, nullptr, 'f'},
	                              {"founders-format", required_argument, nullptr, 1000}, {"min-segment-length", required_argument, nullptr, 'l'},
	                              {"single-threaded", no_argument, nullptr, 1001}, {"help", no_argument, nullptr, 'h'}, {nullptr, 0, nullptr, 0}};
	for (int c; (c = getopt_long(argc, argv, "s:f:l:h", opts, nullptr)) != -1;)
		switch (c)
		{
			case 's': sequences = optarg; break;
			case 'f': founders_path = optarg; break;
			case 'l': min_len = atol(optarg); break;
			case 1000:
				if (0 == strcmp(optarg, "FASTA")) format = 0;
				else if (0 == strcmp(optarg, "text")) format = 1;
				else if (0 == strcmp(optarg, "list-file")) format = 2;
				else { std::cerr << argv[0] << ": invalid argument, \"" << optarg << "\", for option `--founders-format'" << std::endl; return EXIT_FAILURE; }
				break;
			case 1001: single_threaded = true; break;
			case 'h': std::cout << "Usage: match-sequences-to-founders --sequences-list=sequences-list.txt --founders-list=founders.txt\n"; return EXIT_SUCCESS;
			default: return EXIT_FAILURE;
		}
	if (!sequences) { std::cerr << argv[0] << ": '--sequences' ('-s') option required" << std::endl; return EXIT_FAILURE; }
	if (!founders_path) { std::cerr << argv[0] << ": '--founders' ('-f') option required" << std::endl; return EXIT_FAILURE; }
	if (min_len < 0) { std::cerr << "Minimum segment length must be non-negative." << std::endl; return EXIT_FAILURE; }   // main.cc:38-42

	std::cerr << "Reading sequence paths…" << std::endl;
	std::vector<std::string> paths;
	{
		std::ifstream f(sequences);
		if (!f) { std::cerr << "Unable to open the sequence list '" << sequences << "'." << std::endl; return EXIT_FAILURE; }
		aux::read_lines(f, paths);
	}
	std::cerr << "Reading founders…" << std::endl;
	std::vector<std::string> founders;
	if (!read_founders(founders_path, format, founders)) { std::cerr << "Unable to read the founders from '" << founders_path << "'." << std::endl; return EXIT_FAILURE; }
	std::cerr << "Matching founders with sequences…" << std::endl;

	std::ios_base::sync_with_stdio(false);
	std::cout << "SEQUENCE_INDEX" "\t" "LB" "\t" "RB" "\t" "FOUNDER_INDICES" "\n";
	size_t const min_segment_length = (size_t) min_len;
	auto compare = [&](std::vector<size_t> const &src, char c, size_t pos, std::vector<size_t> &dst) {   // match_founder_sequences.cc:108-130
		dst.clear();
		for (size_t const f : src)
			if (pos < founders[f].size() && founders[f][pos] == c) dst.push_back(f);
		return dst.size();
	};
	// One task per input sequence as in match_context::match (match_founder_sequences.cc:218-252): the sequences are matched
	// side by side on the host's threads, one of them reading from the disk at a time (the reference's reading semaphore,
	// :232-241).  A task writes into buffers of its own; the reports leave in input order whatever order the tasks finish in
	// (the reference prints under a lock in completion order: input order is one of those orders, and the only one
	// --single-threaded gives).
	struct Report { std::string out, err; bool unreadable = false; };
	std::vector<Report> reports(paths.size());
	auto match_sequence_and_report = [&](std::string const &sequence, size_t seq_idx, Report &rep) {   // match_founder_sequences.cc:163-218
		std::ostringstream out, err;
		auto output_range = [&](size_t lb, size_t rb, std::vector<size_t> const &idx) {
			out << seq_idx << '\t' << lb << '\t' << rb << '\t';
			for (size_t i = 0; i < idx.size(); ++i) { if (i) out << ','; out << idx[i]; }
			out << '\n';
		};
		auto not_found = [&](char c, size_t pos) {
			err << "Error: character '" << c << "' (" << +c << ") at " << seq_idx << ':' << pos << " not found in the founders.\n";
		};
		std::vector<size_t> cur(founders.size()), dst;
		std::iota(cur.begin(), cur.end(), 0);
		size_t lb = 0, count = cur.size(), dst_count = 0, chr_idx = 0;
		for (char const c : sequence)
		{
			bool recheck = false;
			if (0 != min_segment_length && min_segment_length <= chr_idx - lb) recheck = true;
			else
			{
				dst_count = compare(cur, c, chr_idx, dst);
				if (0 == dst_count)
				{
					if (0 != min_segment_length && chr_idx - lb < min_segment_length)
						err << "Error: segment length " << (lb - chr_idx) << " for sequence " << seq_idx << ':' << chr_idx << " under the given limit.\n";   // (sic: lb - chr_idx, :181)
					recheck = true;
				}
			}
			if (recheck)
			{
				output_range(lb, chr_idx, cur);
				lb = chr_idx;
				cur.resize(founders.size());
				std::iota(cur.begin(), cur.end(), 0);
				dst_count = compare(cur, c, chr_idx, dst);
				if (0 == dst_count) not_found(c, chr_idx);
			}
			std::swap(count, dst_count);
			std::swap(cur, dst);
			++chr_idx;
		}
		if (0 == count) { if (chr_idx) not_found(sequence[chr_idx - 1], chr_idx); }
		else output_range(lb, chr_idx, cur);
		rep.out = out.str();
		rep.err = err.str();
	};
	std::atomic<size_t> next{0};
	std::mutex reading;
	auto worker = [&] {
		for (size_t seq_idx; (seq_idx = next.fetch_add(1)) < paths.size();)
		{
			std::string sequence;
			bool ok;
			{ std::lock_guard<std::mutex> one_reader(reading); ok = aux::read_file(paths[seq_idx], sequence); }
			if (!ok) { reports[seq_idx].unreadable = true; continue; }
			match_sequence_and_report(sequence, seq_idx, reports[seq_idx]);
		}
	};
	size_t n_threads = single_threaded ? 1 : std::max<size_t>(1, std::thread::hardware_concurrency());
	n_threads = std::min(n_threads, std::max<size_t>(1, paths.size()));
	if (1 == n_threads) worker();
	else
	{
		std::vector<std::thread> pool;
		for (size_t t = 0; t < n_threads; ++t) pool.emplace_back(worker);
		for (auto &t : pool) t.join();
	}
	for (size_t seq_idx = 0; seq_idx < paths.size(); ++seq_idx)
	{
		if (reports[seq_idx].unreadable) { std::cout << std::flush; std::cerr << "Unable to open '" << paths[seq_idx] << "'." << std::endl; return EXIT_FAILURE; }
		std::cerr << reports[seq_idx].err;
		std::cout << reports[seq_idx].out;
	}
	std::cout << std::flush;
	return EXIT_SUCCESS;
}
