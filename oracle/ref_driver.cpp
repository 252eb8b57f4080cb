// oracle/ref_driver.cpp -- TEST INFRASTRUCTURE ONLY.
//
// Our own driver around the *reference itself* (d-justen/duckdb-polr compiled from its sources by
// oracle/ref_build.mk into oracle/_ref/libduckdb_ref.so).  It only uses the reference's public C++
// API (duckdb.hpp: DuckDB, Connection, Appender, MaterializedQueryResult).  Used to
//   (1) produce golden vectors for the oracle restatement and the HIP path (tests/golden/, made
//       by tests/golden/make_golden.py), and
//   (2) time the reference's CPU POLAR path as bench.py's cpu_baseline ("kind": "reference").
//
// Script language (one command per line, '#' comments):
//   table <name> <nrows>                       start a table definition
//   col <name> VARCHAR <file>                    strings, each a little-endian u32 length followed by its bytes
//   col <name> <SQLTYPE> <file> [pk]            raw little-endian column file (i32/u32/i64/u16/...); pk: the column is the
//                                               table's PRIMARY KEY (what benchmark/ssb-skew/init/load.sql declares for the
//                                               dimension keys; SelSampleEnumeration reads the constraint)
//   endtable                                    CREATE TABLE + append rows
//   sql <statement>                             run, fail loudly on error
//   query <tag> <statement>                     run, write rows to <outdir>/<tag>.csv, print timing
//   repeat <n> <tag> <statement>                run n times, print wall ms of each run
// The reference writes its POLAR logs to <cwd>/tmp/ (polar_pipeline_executor.cpp:87-106,
// pipeline.cpp:247-263), so the driver chdirs into <outdir> and creates <outdir>/tmp first.
#include "duckdb.hpp"
#include "duckdb/main/appender.hpp"

#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <iostream>
#include <sstream>
#include <string>
#include <sys/stat.h>
#include <unistd.h>
#include <vector>

using namespace duckdb;

struct ColDef {
	std::string name, type, file;
	bool pk = false;
	std::vector<char> data;
	std::vector<size_t> offsets; // VARCHAR: where each row's {length, bytes} record starts
	size_t width;
};

static size_t TypeWidth(const std::string &t) {
	if (t == "TINYINT" || t == "UTINYINT" || t == "BOOLEAN") {
		return 1;
	}
	if (t == "SMALLINT" || t == "USMALLINT") {
		return 2;
	}
	if (t == "INTEGER" || t == "UINTEGER") {
		return 4;
	}
	if (t == "BIGINT" || t == "UBIGINT") {
		return 8;
	}
	fprintf(stderr, "ref_driver: unsupported column type %s\n", t.c_str());
	exit(2);
}

static void Fail(const std::string &what, const std::string &err) {
	fprintf(stderr, "ref_driver: %s failed: %s\n", what.c_str(), err.c_str());
	exit(3);
}

static void LoadTable(Connection &con, const std::string &name, idx_t nrows, std::vector<ColDef> &cols) {
	std::string ddl = "CREATE TABLE " + name + " (";
	for (size_t i = 0; i < cols.size(); i++) {
		ddl += (i ? ", " : "") + cols[i].name + " " + cols[i].type + (cols[i].pk ? " NOT NULL PRIMARY KEY" : "");
	}
	ddl += ")";
	auto r = con.Query(ddl);
	if (r->HasError()) {
		Fail(ddl, r->GetError());
	}
	for (auto &c : cols) {
		std::ifstream f(c.file, std::ios::binary);
		if (!f) {
			Fail("open " + c.file, "cannot open");
		}
		if (c.type == "VARCHAR") {
			c.width = 0;
			c.data.assign(std::istreambuf_iterator<char>(f), std::istreambuf_iterator<char>());
			c.offsets.clear();
			size_t at = 0;
			for (idx_t row = 0; row < nrows; row++) {
				if (at + 4 > c.data.size()) {
					Fail("read " + c.file, "short file");
				}
				uint32_t len;
				memcpy(&len, c.data.data() + at, 4);
				c.offsets.push_back(at);
				at += 4 + len;
			}
			if (at > c.data.size()) {
				Fail("read " + c.file, "short file");
			}
			continue;
		}
		c.width = TypeWidth(c.type);
		c.data.resize(nrows * c.width);
		f.read(c.data.data(), (std::streamsize)c.data.size());
		if ((idx_t)f.gcount() != nrows * c.width) {
			Fail("read " + c.file, "short file");
		}
	}
	Appender appender(con, name);
	for (idx_t row = 0; row < nrows; row++) {
		appender.BeginRow();
		for (auto &c : cols) {
			if (c.type == "VARCHAR") {
				uint32_t len;
				memcpy(&len, c.data.data() + c.offsets[row], 4);
				appender.Append(Value(std::string(c.data.data() + c.offsets[row] + 4, len)));
				continue;
			}
			const char *p = c.data.data() + row * c.width;
			if (c.type == "INTEGER") {
				appender.Append<int32_t>(*(const int32_t *)p);
			} else if (c.type == "UINTEGER") {
				appender.Append<uint32_t>(*(const uint32_t *)p);
			} else if (c.type == "BIGINT") {
				appender.Append<int64_t>(*(const int64_t *)p);
			} else if (c.type == "UBIGINT") {
				appender.Append<uint64_t>(*(const uint64_t *)p);
			} else if (c.type == "SMALLINT") {
				appender.Append<int16_t>(*(const int16_t *)p);
			} else if (c.type == "USMALLINT") {
				appender.Append<uint16_t>(*(const uint16_t *)p);
			} else if (c.type == "TINYINT") {
				appender.Append<int8_t>(*(const int8_t *)p);
			} else if (c.type == "UTINYINT") {
				appender.Append<uint8_t>(*(const uint8_t *)p);
			} else {
				appender.Append<bool>(*(const uint8_t *)p != 0);
			}
		}
		appender.EndRow();
	}
	appender.Close();
	for (auto &c : cols) {
		std::vector<char>().swap(c.data);
	}
}

static void WriteResult(MaterializedQueryResult &res, const std::string &path) {
	std::ofstream out(path);
	for (idx_t c = 0; c < res.ColumnCount(); c++) {
		out << (c ? "," : "") << res.names[c];
	}
	out << "\n";
	for (idx_t r = 0; r < res.RowCount(); r++) {
		for (idx_t c = 0; c < res.ColumnCount(); c++) {
			out << (c ? "," : "") << res.GetValue(c, r).ToString();
		}
		out << "\n";
	}
}

int main(int argc, char **argv) {
	if (argc < 3) {
		fprintf(stderr, "usage: ref_driver <script> <outdir>\n");
		return 1;
	}
	std::ifstream script(argv[1]);
	if (!script) {
		fprintf(stderr, "ref_driver: cannot open %s\n", argv[1]);
		return 1;
	}
	std::string outdir = argv[2];
	mkdir(outdir.c_str(), 0755);
	mkdir((outdir + "/tmp").c_str(), 0755);
	if (chdir(outdir.c_str()) != 0) {
		perror("chdir");
		return 1;
	}

	// POLR_REF_OPEN_THREADS=<n>: open the database with at most n worker threads (DBConfig::maximum_threads) instead of one
	// per hardware thread.  (On the 256-thread hosts of the GPU boxes the reference's GROUP BY plans end in "Invalid Error:
	// vector::reserve" when the instance was OPENED with 256 threads, whatever `SET threads` says afterwards.)
	DBConfig config;
	if (const char *open_threads = getenv("POLR_REF_OPEN_THREADS")) {
		const long n = atol(open_threads);
		if (n > 0) {
			config.options.maximum_threads = (idx_t)n;
		}
	}
	DuckDB db(nullptr, &config);
	Connection con(db);

	std::string line, tname;
	idx_t tnrows = 0;
	std::vector<ColDef> tcols;
	while (std::getline(script, line)) {
		if (line.empty() || line[0] == '#') {
			continue;
		}
		std::istringstream ss(line);
		std::string cmd;
		ss >> cmd;
		if (cmd == "table") {
			ss >> tname >> tnrows;
			tcols.clear();
		} else if (cmd == "col") {
			ColDef c;
			std::string flag;
			ss >> c.name >> c.type >> c.file >> flag;
			c.pk = flag == "pk";
			tcols.push_back(c);
		} else if (cmd == "endtable") {
			auto t0 = std::chrono::steady_clock::now();
			LoadTable(con, tname, tnrows, tcols);
			double ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
			printf("loaded %s rows=%llu ms=%.1f\n", tname.c_str(), (unsigned long long)tnrows, ms);
		} else if (cmd == "sql") {
			std::string stmt;
			std::getline(ss, stmt);
			auto r = con.Query(stmt);
			if (r->HasError()) {
				Fail(stmt, r->GetError());
			}
		} else if (cmd == "query" || cmd == "repeat") {
			idx_t n = 1;
			if (cmd == "repeat") {
				ss >> n;
			}
			std::string tag, stmt;
			ss >> tag;
			std::getline(ss, stmt);
			for (idx_t i = 0; i < n; i++) {
				auto t0 = std::chrono::steady_clock::now();
				auto r = con.Query(stmt);
				double ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
				if (r->HasError()) {
					Fail(stmt, r->GetError());
				}
				printf("query %s run=%llu rows=%llu wall_ms=%.3f\n", tag.c_str(), (unsigned long long)i,
				       (unsigned long long)r->RowCount(), ms);
				if (i == 0) {
					WriteResult(*r, tag + ".csv");
				}
			}
		} else {
			fprintf(stderr, "ref_driver: unknown command '%s'\n", cmd.c_str());
			return 2;
		}
		fflush(stdout);
	}
	return 0;
}
