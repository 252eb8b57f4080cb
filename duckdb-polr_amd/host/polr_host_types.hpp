// duckdb-polr_amd/host/polr_host_types.hpp -- the boundary types of the host mirror.
//
// Same names and meaning as the reference's types on the POLAR path, cut down
// to what the path touches: flat / dictionary vectors of fixed-width cells, selection vectors,
// DataChunk with Reference/Slice/Reset, OperatorResultType, ExecutionContext with the POLAR knobs
// of ClientConfig and the thread-local current_join_path.
//   DataChunk           src/include/duckdb/common/types/data_chunk.hpp:43-160
//   Vector              src/include/duckdb/common/types/vector.hpp:36-140
//   SelectionVector     src/include/duckdb/common/types/selection_vector.hpp
//   OperatorResultType  src/include/duckdb/common/enums/operator_result_type.hpp:24
//   ClientConfig knobs  src/include/duckdb/main/client_config.hpp:76-93, main/config.hpp:41-50,141-144
#pragma once

#include <cstdint>
#include <cstring>
#include <map>
#include <memory>
#include <stdexcept>
#include <string>
#include <vector>

namespace duckdb_polr {

typedef uint64_t idx_t;
typedef uint32_t sel_t;
#ifndef POLR_STANDARD_VECTOR_SIZE
#define POLR_STANDARD_VECTOR_SIZE 1024 // vector_size.hpp:16-18 of the reference snapshot
#endif
static const idx_t STANDARD_VECTOR_SIZE = POLR_STANDARD_VECTOR_SIZE;

using std::string;
using std::unique_ptr;
using std::vector;

struct InternalException : public std::runtime_error {
	explicit InternalException(const string &msg) : std::runtime_error(msg) {
	}
};
struct NotImplementedException : public std::runtime_error {
	explicit NotImplementedException(const string &msg) : std::runtime_error(msg) {
	}
};

enum class OperatorResultType : uint8_t { NEED_MORE_INPUT, HAVE_MORE_OUTPUT, FINISHED };

enum class PhysicalOperatorType : uint8_t { HASH_JOIN, MULTIPLEXER, ADAPTIVE_UNION, TABLE_SCAN };

enum class JoinType : uint8_t { INNER, LEFT, RIGHT, SEMI, ANTI, MARK, SINGLE, OUTER };

// same order as src/include/duckdb/main/config.hpp:41-50
enum class MultiplexerRouting : uint8_t {
	ALTERNATE,
	ADAPTIVE_REINIT,
	DYNAMIC,
	INIT_ONCE,
	OPPORTUNISTIC,
	DEFAULT_PATH,
	BACKPRESSURE,
	EXPONENTIAL_BACKOFF
};

// same order as src/include/duckdb/common/enums/join_enumerator.hpp:15-25
enum class JoinEnumerator : uint8_t {
	DFS_RANDOM,
	DFS_MIN_CARD,
	DFS_UNCERTAIN,
	BFS_RANDOM,
	BFS_MIN_CARD,
	BFS_UNCERTAIN,
	EACH_LAST_ONCE,
	EACH_FIRST_ONCE,
	SAMPLE
};

// fixed-width physical types on the path (keys: integers; payload: any fixed width incl. the 16-byte
// string_t cell, row_gather.cpp:47-86)
struct LogicalType {
	uint32_t width = 4;
	bool is_signed = true;
	LogicalType() {
	}
	LogicalType(uint32_t w, bool s) : width(w), is_signed(s) {
	}
	bool operator==(const LogicalType &o) const {
		return width == o.width && is_signed == o.is_signed;
	}
	static LogicalType INTEGER() {
		return LogicalType(4, true);
	}
	static LogicalType UINTEGER() {
		return LogicalType(4, false);
	}
	static LogicalType BIGINT() {
		return LogicalType(8, true);
	}
	static LogicalType USMALLINT() {
		return LogicalType(2, false);
	}
	static LogicalType VARCHAR() {
		return LogicalType(16, false);
	}
};

struct SelectionVector {
	std::shared_ptr<vector<sel_t>> owned;
	sel_t *sel_vector = nullptr;
	void Initialize(idx_t count = STANDARD_VECTOR_SIZE) {
		owned = std::make_shared<vector<sel_t>>(count);
		sel_vector = owned->data();
	}
	sel_t *data() {
		return sel_vector;
	}
	idx_t get_index(idx_t i) const {
		return sel_vector ? sel_vector[i] : i;
	}
	void set_index(idx_t i, idx_t v) {
		sel_vector[i] = (sel_t)v;
	}
};

// FLAT (sel == nullptr) or DICTIONARY (sel over the base buffer) vector; buffers are shared, never
// copied by Reference/Slice -- the zero-copy behaviour the reference relies on
// (routing_strategy.cpp:18-24, join_hashtable.cpp:555).
struct Vector {
	LogicalType type;
	std::shared_ptr<vector<uint8_t>> buffer;   // owns `data` when allocated here
	std::shared_ptr<vector<uint8_t>> vbuffer;  // owns `validity`
	uint8_t *data = nullptr;
	uint8_t *validity = nullptr; // nullptr = all valid, else one byte per base row
	SelectionVector sel;         // dictionary selection (sel.sel_vector == nullptr: flat)

	Vector() {
	}
	explicit Vector(LogicalType t, idx_t capacity = STANDARD_VECTOR_SIZE) : type(t) {
		buffer = std::make_shared<vector<uint8_t>>(capacity * t.width);
		data = buffer->data();
	}
	void Reference(const Vector &other) {
		*this = other;
	}
	void Slice(const SelectionVector &s, idx_t count) {
		// compose selections: new_sel[i] = old_sel[s[i]]
		SelectionVector ns;
		ns.Initialize(count);
		for (idx_t i = 0; i < count; i++) {
			ns.set_index(i, sel.get_index(s.get_index(i)));
		}
		sel = ns;
	}
	const uint8_t *Cell(idx_t i) const {
		return data + sel.get_index(i) * (idx_t)type.width;
	}
	bool IsValid(idx_t i) const {
		return !validity || validity[sel.get_index(i)] != 0;
	}
	void EnsureValidity(idx_t capacity) {
		if (!validity) {
			vbuffer = std::make_shared<vector<uint8_t>>(capacity, (uint8_t)1);
			validity = vbuffer->data();
		}
	}
};

class DataChunk {
public:
	vector<Vector> data;
	void Initialize(const vector<LogicalType> &types) {
		data.clear();
		for (auto &t : types) {
			data.emplace_back(t);
		}
		count = 0;
		initial_types = types;
	}
	void InitializeEmpty(const vector<LogicalType> &types) {
		data.assign(types.size(), Vector());
		for (idx_t i = 0; i < types.size(); i++) {
			data[i].type = types[i];
		}
		count = 0;
		initial_types = types;
	}
	idx_t size() const {
		return count;
	}
	idx_t ColumnCount() const {
		return data.size();
	}
	void SetCardinality(idx_t c) {
		count = c;
	}
	void SetCardinality(const DataChunk &o) {
		count = o.count;
	}
	vector<LogicalType> GetTypes() const {
		vector<LogicalType> t;
		for (auto &v : data) {
			t.push_back(v.type);
		}
		return t;
	}
	void Reference(DataChunk &other) {
		for (idx_t i = 0; i < other.ColumnCount() && i < ColumnCount(); i++) {
			data[i].Reference(other.data[i]);
		}
		count = other.count;
	}
	// DataChunk::Slice(other, sel, count): columns become dictionary vectors over other's buffers
	void Slice(DataChunk &other, const SelectionVector &sel, idx_t count_p, idx_t col_offset = 0) {
		for (idx_t i = 0; i < other.ColumnCount(); i++) {
			data[col_offset + i].Reference(other.data[i]);
			data[col_offset + i].Slice(sel, count_p);
		}
		count = count_p;
	}
	void Reset() {
		// fresh buffers so that vectors referenced elsewhere stay intact (DataChunk::Reset re-points the
		// vectors at the chunk's own buffers in the reference)
		Initialize(initial_types);
	}
	void Verify() const {
	}

private:
	idx_t count = 0;
	vector<LogicalType> initial_types;
};

// the POLAR knobs (client_config.hpp:76-93, config.hpp:141-144); set through SET/PRAGMA in the
// reference (settings.cpp:655-817, pragma_functions.cpp:118-219)
struct ClientConfig {
	bool enable_polr = false;
	bool bushy_polr = false;
	bool log_tuples_routed = false;
	bool measure_polr_pipeline = false;
	bool caching = true;
	bool lip = false;
	bool time_resistance = false;
	JoinEnumerator join_enumerator = JoinEnumerator::SAMPLE;
	idx_t max_join_orders = 8;
	idx_t init_tuple_count = 1024;
	idx_t atc_multiplier = 1;
	// DBConfig options
	double regret_budget = 0.01;
	MultiplexerRouting multiplexer_routing = MultiplexerRouting::ADAPTIVE_REINIT;
	string dir_prefix;
	idx_t threads = 1;
};

struct ClientContext {
	ClientConfig config;
	bool interrupted = false;
};

struct ThreadContext {
	vector<idx_t> *current_join_path = nullptr; // thread_context.hpp:26
};

struct ExecutionContext {
	ClientContext &client;
	ThreadContext &thread;
	ExecutionContext(ClientContext &c, ThreadContext &t) : client(c), thread(t) {
	}
};

class PhysicalOperator;

class OperatorState {
public:
	virtual ~OperatorState() {
	}
	virtual void Finalize(PhysicalOperator *op, ExecutionContext &context) {
	}
};

class GlobalOperatorState {
public:
	virtual ~GlobalOperatorState() {
	}
};

// physical_operator.hpp:85-141
class PhysicalOperator {
public:
	PhysicalOperator(PhysicalOperatorType type_p, vector<LogicalType> types_p, idx_t estimated_cardinality_p)
	    : type(type_p), types(std::move(types_p)), estimated_cardinality(estimated_cardinality_p) {
		op_state.reset(new GlobalOperatorState());
	}
	virtual ~PhysicalOperator() {
	}
	PhysicalOperatorType type;
	vector<LogicalType> types;
	idx_t estimated_cardinality;
	unique_ptr<GlobalOperatorState> op_state;

	const vector<LogicalType> &GetTypes() const {
		return types;
	}
	virtual unique_ptr<OperatorState> GetOperatorState(ExecutionContext &context) const {
		return unique_ptr<OperatorState>(new OperatorState());
	}
	virtual OperatorResultType Execute(ExecutionContext &context, DataChunk &input, DataChunk &chunk,
	                                   GlobalOperatorState &gstate, OperatorState &state) const = 0;
	virtual bool ParallelOperator() const {
		return false;
	}
	virtual bool RequiresCache() const {
		return false;
	}
	virtual string ParamsToString() const {
		return "";
	}
};

} // namespace duckdb_polr
