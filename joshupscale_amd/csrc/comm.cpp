// Multi-GPU start-up in the C layer: the RCCL broadcast of the model container.
//
// The reference has no distributed code at all; BASELINE.json
// config 4 adds N independent streams on N GPUs with ONE collective: the model bytes go
// from rank 0 to every rank over xGMI, so that only rank 0 touches the file system.
// One process per GPU; the launcher (bench.py under torch.distributed.run, or any other)
// carries the 128-byte communicator id from rank 0 to the other ranks.
//
// librccl is opened at first use (dlopen), not linked: a single-GPU caller of
// libJoshUpscale.so -- the AviSynth/OBS plugins -- never loads it.
#include <dlfcn.h>
#include <rccl/rccl.h>

#include <cstring>
#include <mutex>
#include <stdexcept>
#include <string>

#include "hip_util.h"
#include "joshupscale_amd.h"

namespace ju {

namespace {

struct Rccl {
	void *lib = nullptr;
	decltype(&ncclGetUniqueId) getUniqueId = nullptr;
	decltype(&ncclCommInitRank) commInitRank = nullptr;
	decltype(&ncclBroadcast) broadcast = nullptr;
	decltype(&ncclAllReduce) allReduce = nullptr;
	decltype(&ncclCommCount) commCount = nullptr;
	decltype(&ncclCommDestroy) commDestroy = nullptr;
	decltype(&ncclCommAbort) commAbort = nullptr;
	decltype(&ncclGetErrorString) errorString = nullptr;
};

const Rccl &rccl() {
	static Rccl r;
	static std::once_flag once;
	static std::string failure;
	std::call_once(once, [] {
		// a process that already initialised torch.distributed's "nccl" backend has an
		// RCCL loaded: the SONAME lookup finds that one first
		for (const char *name : {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"}) {
			r.lib = dlopen(name, RTLD_NOW | RTLD_GLOBAL);
			if (r.lib) break;
		}
		if (!r.lib) {
			failure = std::string("cannot load librccl: ") + dlerror();
			return;
		}
		auto sym = [&](const char *n) -> void * {
			void *p = dlsym(r.lib, n);
			if (!p && failure.empty()) failure = std::string("librccl lacks ") + n;
			return p;
		};
		r.getUniqueId = reinterpret_cast<decltype(r.getUniqueId)>(sym("ncclGetUniqueId"));
		r.commInitRank = reinterpret_cast<decltype(r.commInitRank)>(sym("ncclCommInitRank"));
		r.broadcast = reinterpret_cast<decltype(r.broadcast)>(sym("ncclBroadcast"));
		r.allReduce = reinterpret_cast<decltype(r.allReduce)>(sym("ncclAllReduce"));
		r.commCount = reinterpret_cast<decltype(r.commCount)>(sym("ncclCommCount"));
		r.commDestroy = reinterpret_cast<decltype(r.commDestroy)>(sym("ncclCommDestroy"));
		// optional.  An RCCL without it has no way to tear down a communicator whose peers may be stuck in a
		// collective (ncclCommDestroy may wait for outstanding operations): the failure path then LEAKS the
		// communicator and throws, so that this rank exits non-zero and the launcher kills the job (advisor, round 4)
		r.commAbort = reinterpret_cast<decltype(r.commAbort)>(dlsym(r.lib, "ncclCommAbort"));
		r.errorString = reinterpret_cast<decltype(r.errorString)>(sym("ncclGetErrorString"));
	});
	if (!failure.empty()) throw std::runtime_error(failure);
	return r;
}

void ncclCheck(ncclResult_t e, const char *what) {
	if (e != ncclSuccess) {
		throw std::runtime_error(std::string(what) + ": " + rccl().errorString(e));
	}
}

}  // namespace

}  // namespace ju

struct ju_comm {
	ncclComm_t comm = nullptr;
	int device = 0, rank = 0, world = 1;
	hipStream_t stream = nullptr;
};

// (c_api.cpp: maps exceptions to JU_ERR_* and keeps the message for ju_last_error)
int juGuarded(void (*fn)(void *), void *ctx);

namespace {

template <typename F>
int guardedCall(F &&f) {
	return juGuarded([](void *c) { (*static_cast<F *>(c))(); }, &f);
}

static_assert(JU_COMM_ID_BYTES == NCCL_UNIQUE_ID_BYTES, "ju_comm id size must equal RCCL's");

}  // namespace

extern "C" {

int ju_comm_unique_id(void *id_out) {
	return guardedCall([&] {
		if (id_out == nullptr) throw std::invalid_argument("ju_comm_unique_id: id_out is NULL");
		ncclUniqueId id;
		ju::ncclCheck(ju::rccl().getUniqueId(&id), "ncclGetUniqueId");
		std::memcpy(id_out, &id, sizeof(id));
	});
}

int ju_comm_create(const void *id, int rank, int world_size, int device_id, ju_comm **out_comm) {
	return guardedCall([&] {
		if (id == nullptr || out_comm == nullptr) throw std::invalid_argument("ju_comm_create: NULL argument");
		*out_comm = nullptr;
		if (world_size < 1 || rank < 0 || rank >= world_size) {
			throw std::invalid_argument("ju_comm_create: rank must be in [0, world_size)");
		}
		int count = 0;
		JU_HIP(hipGetDeviceCount(&count));
		if (device_id < 0 || device_id >= count) throw std::invalid_argument("ju_comm_create: no such device");
		ju::DeviceGuard guard(device_id);
		auto c = new ju_comm();
		c->device = device_id;
		c->rank = rank;
		c->world = world_size;
		try {
			ncclUniqueId uid;
			std::memcpy(&uid, id, sizeof(uid));
			ju::ncclCheck(ju::rccl().commInitRank(&c->comm, world_size, uid, rank), "ncclCommInitRank");
			JU_HIP(hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking));
		} catch (...) {
			if (c->comm) (void)ju::rccl().commDestroy(c->comm);
			delete c;
			throw;
		}
		*out_comm = c;
	});
}

int ju_comm_broadcast(ju_comm *comm, void *bytes, size_t size, int root) {
	return guardedCall([&] {
		if (comm == nullptr || (bytes == nullptr && size != 0)) throw std::invalid_argument("ju_comm_broadcast: NULL argument");
		if (comm->comm == nullptr) throw std::runtime_error("ju_comm_broadcast: the communicator was aborted by an earlier failure");
		if (root < 0 || root >= comm->world) throw std::invalid_argument("ju_comm_broadcast: bad root");
		ju::DeviceGuard guard(comm->device);
		// Everything that can fail on THIS rank alone happens before the payload collective, so that
		// a local failure is at least never left half-way into it.  On such a failure this rank's
		// communicator is aborted (ncclCommAbort tears down THIS rank only): later calls on it fail
		// at once instead of entering a collective its peers have given up on.  The peers are NOT
		// guaranteed to see an error -- a rank blocked in the collective stays blocked until its own
		// watchdog or the launcher's timeout ends it (torch.distributed.run kills the job when one
		// rank exits non-zero, which is what bench.py relies on).
		ju::DeviceBuffer dev, agree;
		try {
			dev = ju::DeviceBuffer(size ? size : 1);
			agree = ju::DeviceBuffer(2 * sizeof(long long));
			if (comm->rank == root && size) {
				JU_HIP(hipMemcpyAsync(dev.get(), bytes, size, hipMemcpyHostToDevice, comm->stream));
				JU_HIP(hipStreamSynchronize(comm->stream));
			}
		} catch (...) {
			if (ju::rccl().commAbort) (void)ju::rccl().commAbort(comm->comm);  // (absent: leaked, see rccl())
			comm->comm = nullptr;
			throw;
		}
		// the ranks' sizes must agree: max over ranks of (size, -size) -- every rank sees the
		// same two numbers and takes the same decision
		long long sz[2] = {static_cast<long long>(size), -static_cast<long long>(size)};
		JU_HIP(hipMemcpyAsync(agree.get(), sz, sizeof(sz), hipMemcpyHostToDevice, comm->stream));
		ju::ncclCheck(ju::rccl().allReduce(agree.get(), agree.get(), 2, ncclInt64, ncclMax, comm->comm, comm->stream),
		    "ncclAllReduce");
		JU_HIP(hipMemcpyAsync(sz, agree.get(), sizeof(sz), hipMemcpyDeviceToHost, comm->stream));
		JU_HIP(hipStreamSynchronize(comm->stream));
		if (sz[0] != -sz[1]) {
			throw std::invalid_argument("ju_comm_broadcast: the ranks passed different sizes (" + std::to_string(-sz[1]) +
			                            " .. " + std::to_string(sz[0]) + " bytes)");
		}
		if (size == 0) return;
		// device (root) -> ncclBroadcast of uint8 over xGMI -> host (others)
		ju::ncclCheck(ju::rccl().broadcast(dev.get(), dev.get(), size, ncclUint8, root, comm->comm, comm->stream),
		    "ncclBroadcast");
		if (comm->rank != root) JU_HIP(hipMemcpyAsync(bytes, dev.get(), size, hipMemcpyDeviceToHost, comm->stream));
		JU_HIP(hipStreamSynchronize(comm->stream));
	});
}

int ju_comm_allreduce_max(ju_comm *comm, double *value) {
	return guardedCall([&] {
		if (comm == nullptr || value == nullptr) throw std::invalid_argument("ju_comm_allreduce_max: NULL argument");
		if (comm->comm == nullptr) throw std::runtime_error("ju_comm_allreduce_max: the communicator was aborted");
		ju::DeviceGuard guard(comm->device);
		ju::DeviceBuffer dev;
		try {
			dev = ju::DeviceBuffer(sizeof(double));
		} catch (...) {  // (peers must not wait for this rank for ever)
			if (ju::rccl().commAbort) (void)ju::rccl().commAbort(comm->comm);  // (absent: leaked, see rccl())
			comm->comm = nullptr;
			throw;
		}
		JU_HIP(hipMemcpyAsync(dev.get(), value, sizeof(double), hipMemcpyHostToDevice, comm->stream));
		ju::ncclCheck(ju::rccl().allReduce(dev.get(), dev.get(), 1, ncclFloat64, ncclMax, comm->comm, comm->stream),
		    "ncclAllReduce");
		JU_HIP(hipMemcpyAsync(value, dev.get(), sizeof(double), hipMemcpyDeviceToHost, comm->stream));
		JU_HIP(hipStreamSynchronize(comm->stream));
	});
}

int ju_comm_count(const ju_comm *comm, int *count) {
	return guardedCall([&] {
		if (comm == nullptr || count == nullptr) throw std::invalid_argument("ju_comm_count: NULL argument");
		if (comm->comm == nullptr) throw std::runtime_error("ju_comm_count: the communicator was aborted");
		ju::ncclCheck(ju::rccl().commCount(comm->comm, count), "ncclCommCount");
	});
}

void ju_comm_destroy(ju_comm *comm) {
	if (comm == nullptr) return;
	try {
		ju::DeviceGuard guard(comm->device);
		if (comm->stream) (void)hipStreamDestroy(comm->stream);
		if (comm->comm) (void)ju::rccl().commDestroy(comm->comm);
	} catch (...) {
	}
	delete comm;
}

}  // extern "C"
