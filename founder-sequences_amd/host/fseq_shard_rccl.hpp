// fseq_shard_rccl.hpp -- one alignment over the GPUs of one node from a C++17 host: RCCL bound to the C ABI.
//
// The reference is one process that spreads independent update_pbwt_task's over a concurrent dispatch queue
// (founder-sequences/main.cc:68-150, founder-sequences/segmentation_lp_context.cc:319-332).  Here: one process,
// one host thread and one device per rank; rank r's context is made a shard of the alignment with fseq_set_shard
// (include/fseq.h), whose one exchange primitive -- an in-place all-reduce over the rank's device buffer -- is
// ncclAllReduce(ncclUint32, ncclSum | ncclMax) on a communicator made by ncclCommInitAll, on a stream of its own.
// Every rank makes the same calls in the same order (the threads below run the same function); the results are
// identical on all ranks, boundary states sit on their owners (fseq_shard_owner).
#pragma once

#include <fseq.h>

#include <hip/hip_runtime.h>
#include <rccl/rccl.h>

#include <condition_variable>
#include <cstdint>
#include <functional>
#include <mutex>
#include <string>
#include <thread>
#include <vector>

namespace fseq_host {

struct shard_link {
	ncclComm_t comm = nullptr;
	hipStream_t stream = nullptr;
	std::uint32_t *xbuf = nullptr;
	std::uint64_t words = 0;
	std::uint64_t calls = 0, words_moved = 0;
};

// fseq_allreduce_fn: xbuf[off .. off + count) over all ranks, in place; returns once the result is visible to work
// submitted afterwards on any stream of the device
inline int rccl_allreduce(void *user, std::uint64_t off, std::uint64_t count, int op)
{
	auto *l(static_cast<shard_link *>(user));
	++l->calls; l->words_moved += count;
	if (ncclSuccess != ncclAllReduce(l->xbuf + off, l->xbuf + off, count, ncclUint32, op ? ncclMax : ncclSum, l->comm, l->stream)) return 1;
	return hipSuccess == hipStreamSynchronize(l->stream) ? 0 : 1;
}

// The ranks are threads of one process, so they can agree on the host before a collective: everybody arrives with a
// status word and leaves with the largest one.  A rank that failed before the first exchange (no context, no exchange
// buffer) makes the others return instead of waiting in an all-reduce it will never join.
class agreement {
public:
	explicit agreement(int parties = 1): m_parties(parties) {}
	void reset(int parties) { m_parties = parties; m_waiting = 0; m_acc = 0; m_result = 0; ++m_generation; }
	int arrive(int status)
	{
		std::unique_lock<std::mutex> lock(m_mutex);
		if (status > m_acc) m_acc = status;
		unsigned long const gen(m_generation);
		if (++m_waiting == m_parties)
		{
			m_result = m_acc; m_acc = 0; m_waiting = 0; ++m_generation;
			m_cv.notify_all();
			return m_result;
		}
		m_cv.wait(lock, [&]() { return gen != m_generation; });
		return m_result;
	}
private:
	std::mutex m_mutex;
	std::condition_variable m_cv;
	int m_parties = 1, m_waiting = 0, m_acc = 0, m_result = 0;
	unsigned long m_generation = 0;
};

class rccl_world {
public:
	// devices 0 .. world - 1 of this node
	bool init(int world, std::string &err)
	{
		int ndev = 0;
		if (hipSuccess != hipGetDeviceCount(&ndev) || ndev < world)
		{
			err = "--gpus " + std::to_string(world) + ": the node has " + std::to_string(ndev) + " GPU(s)";
			return false;
		}
		m_links.resize(world);
		m_agree.reset(world);
		std::vector<ncclComm_t> comms(world);
		std::vector<int> devs(world);
		for (int r = 0; r < world; ++r) devs[r] = r;
		ncclResult_t const rc(ncclCommInitAll(comms.data(), world, devs.data()));
		if (ncclSuccess != rc) { err = std::string("ncclCommInitAll: ") + ncclGetErrorString(rc); return false; }
		for (int r = 0; r < world; ++r)
		{
			m_links[r].comm = comms[r];
			if (hipSuccess != hipSetDevice(r) || hipSuccess != hipStreamCreateWithFlags(&m_links[r].stream, hipStreamNonBlocking))
			{ err = "unable to create the exchange stream of rank " + std::to_string(r); return false; }
		}
		return true;
	}

	int world() const { return (int) m_links.size(); }
	// every rank thread calls this with its own status (0 = fine); all leave with the largest one
	int agree(int status) { return m_agree.arrive(status); }
	shard_link &link(int rank) { return m_links[rank]; }

	// the exchange buffer of a rank, sized for its context (fseq_shard_xbuf_words), and the context made a shard
	int attach(fseq_ctx *ctx, int rank)
	{
		shard_link &l(m_links[rank]);
		(void) hipSetDevice(rank);
		l.words = fseq_shard_xbuf_words(ctx, (std::uint32_t) world());
		if (hipSuccess != hipMalloc(reinterpret_cast<void **>(&l.xbuf), l.words * 4)) return FSEQ_E_OOM;
		(void) hipMemset(l.xbuf, 0, l.words * 4);
		return fseq_set_shard(ctx, (std::uint32_t) rank, (std::uint32_t) world(), l.xbuf, l.words, &rccl_allreduce, &l);
	}

	// a one-word all-reduce over all ranks (sum of rank + 1): the transport works before any alignment depends on it
	bool self_test(std::string &err)
	{
		int const W(world());
		std::vector<std::uint32_t *> bufs(W, nullptr);
		std::vector<std::uint32_t> got(W, 0);
		std::vector<std::thread> ths;
		std::vector<int> bad(W, 0);
		for (int r = 0; r < W; ++r)
			ths.emplace_back([&, r]() {
				(void) hipSetDevice(r);
				std::uint32_t const mine(r + 1);
				bool ok(hipSuccess == hipMalloc(reinterpret_cast<void **>(&bufs[r]), 4) && hipSuccess == hipMemcpy(bufs[r], &mine, 4, hipMemcpyHostToDevice));
				// (a rank that could not set up must not leave the others in the collective: agree on the host first)
				if (agree(ok ? 0 : 1) != 0) { bad[r] = ok ? 0 : 1; if (bufs[r]) (void) hipFree(bufs[r]); return; }
				shard_link probe(m_links[r]);
				probe.xbuf = bufs[r];
				if (rccl_allreduce(&probe, 0, 1, 0) != 0) { bad[r] = 1; return; }
				if (hipSuccess != hipMemcpy(&got[r], bufs[r], 4, hipMemcpyDeviceToHost)) bad[r] = 1;
				(void) hipFree(bufs[r]);
			});
		for (auto &t : ths) t.join();
		std::uint32_t const want((std::uint32_t) (W * (W + 1) / 2));
		for (int r = 0; r < W; ++r)
			if (bad[r]) { err = "RCCL self-test: rank " + std::to_string(r) + " could not set up its buffer"; return false; }
		for (int r = 0; r < W; ++r)
			if (got[r] != want) { err = "RCCL self-test failed on rank " + std::to_string(r); return false; }
		return true;
	}

	// fn(rank) on every rank's own thread and device; returns the ranks' return codes
	std::vector<int> run(std::function<int(int)> const &fn)
	{
		int const W(world());
		std::vector<int> rc(W, 0);
		std::vector<std::thread> ths;
		for (int r = 1; r < W; ++r) ths.emplace_back([&, r]() { (void) hipSetDevice(r); rc[r] = fn(r); });
		(void) hipSetDevice(0);
		rc[0] = fn(0);
		for (auto &t : ths) t.join();
		return rc;
	}

	~rccl_world()
	{
		for (size_t r = 0; r < m_links.size(); ++r)
		{
			(void) hipSetDevice((int) r);
			if (m_links[r].xbuf) (void) hipFree(m_links[r].xbuf);
			if (m_links[r].stream) (void) hipStreamDestroy(m_links[r].stream);
			if (m_links[r].comm) (void) ncclCommDestroy(m_links[r].comm);
		}
	}

private:
	std::vector<shard_link> m_links;
	agreement m_agree;
};

} // namespace fseq_host
