# oracle/ref_build.mk -- TEST INFRASTRUCTURE ONLY (see oracle/README.md).
#
# Compiles the upstream reference (d-justen/duckdb-polr, mounted read-only at
# $(REF)) *from its sources where they lie* into oracle/_ref/libduckdb_ref.so,
# plus our own driver oracle/ref_driver.cpp that talks to it through the
# reference's public C++ API (duckdb.hpp).  Nothing is copied out of $(REF),
# nothing is stubbed, the reference's own build system (cmake) is not run:
# this file is a plain glob-and-compile recipe.  Outputs go only to
# oracle/_ref/ (git-ignored, but shipped to the GPU box by gpurun).
#
#   make -f oracle/ref_build.mk -j8        # ~100 unity units + ~60 third-party files
#
REF      ?= /root/reference
OUT      ?= $(dir $(abspath $(lastword $(MAKEFILE_LIST))))_ref
OBJ      := $(OUT)/obj
CXX      ?= g++
OPT      ?= -O2
TP       := $(REF)/third_party

INC := -I$(REF)/src/include -I$(TP)/fsst -I$(TP)/fmt/include -I$(TP)/hyperloglog \
       -I$(TP)/fastpforlib -I$(TP)/fast_float -I$(TP)/re2 -I$(TP)/miniz \
       -I$(TP)/utf8proc/include -I$(TP)/miniparquet -I$(TP)/concurrentqueue \
       -I$(TP)/pcg -I$(TP)/tdigest -I$(TP)/mbedtls/include -I$(TP)/jaro_winkler \
       -I$(TP)/libpg_query/include -I$(TP)/httplib
DEFS := -DDUCKDB -DDUCKDB_MAIN_LIBRARY -DNDEBUG \
        -DDUCKDB_VERSION='"v0.0.1-dev0"' -DDUCKDB_SOURCE_ID='"polr-ref"'
CXXFLAGS := -std=c++11 $(OPT) -fPIC -w -pthread $(DEFS) $(INC)

# The reference's own sources only compile the way its cmake arranges them: per-directory "unity"
# units (several files rely on includes made by the file concatenated before them).  ref_unity.py
# reads the add_library_unity() lists out of the reference's CMakeLists.txt *as data* and writes
# one-line-per-source #include units into $(OUT)/ub/ (the sources stay where they lie).
HERE0    := $(dir $(abspath $(lastword $(MAKEFILE_LIST))))
CORE_SRC := $(shell python3 $(HERE0)ref_unity.py $(REF) $(OUT)/ub)
TP_SRC := \
  $(TP)/fsst/libfsst.cpp $(TP)/fsst/fsst_avx512.cpp \
  $(TP)/fmt/format.cc \
  $(addprefix $(TP)/libpg_query/, postgres_parser.cpp pg_functions.cpp \
     src_backend_parser_parser.cpp src_backend_parser_gram.cpp src_backend_nodes_list.cpp \
     src_backend_parser_scansup.cpp src_backend_nodes_makefuncs.cpp src_backend_nodes_value.cpp \
     src_backend_parser_scan.cpp src_common_keywords.cpp) \
  $(addprefix $(TP)/re2/re2/, bitstate.cc compile.cc dfa.cc filtered_re2.cc mimics_pcre.cc nfa.cc \
     onepass.cc parse.cc perl_groups.cc prefilter.cc prefilter_tree.cc prog.cc re2.cc regexp.cc \
     set.cc simplify.cc stringpiece.cc tostring.cc unicode_casefold.cc unicode_groups.cc) \
  $(TP)/re2/util/rune.cc $(TP)/re2/util/strutil.cc \
  $(TP)/miniz/miniz.cpp \
  $(TP)/utf8proc/utf8proc.cpp $(TP)/utf8proc/utf8proc_wrapper.cpp \
  $(TP)/hyperloglog/hyperloglog.cpp $(TP)/hyperloglog/sds.cpp \
  $(TP)/fastpforlib/bitpacking.cpp \
  $(TP)/mbedtls/mbedtls_wrapper.cpp \
  $(addprefix $(TP)/mbedtls/library/, sha256.cpp rsa.cpp rsa_alt_helpers.cpp md.cpp bignum.cpp \
     oid.cpp constant_time.cpp platform_util.cpp base64.cpp pkparse.cpp pk.cpp pk_wrap.cpp \
     asn1parse.cpp pem.cpp)

ALL_SRC := $(CORE_SRC) $(TP_SRC)
# object name = source path relative to $(REF) with '/' -> '__'
objname = $(OBJ)/$(subst /,__,$(patsubst $(OUT)/%,%,$(patsubst $(REF)/%,%,$(basename $(1))))).o
ALL_OBJ := $(foreach s,$(ALL_SRC),$(call objname,$(s)))

LIB    := $(OUT)/libduckdb_ref.so
DRIVER := $(OUT)/ref_driver
HERE   := $(dir $(abspath $(lastword $(MAKEFILE_LIST))))

.PHONY: all lib driver clean
all: lib driver
lib: $(LIB)
driver: $(DRIVER)

$(OBJ):
	@mkdir -p $(OBJ)

define compile_rule
$(call objname,$(1)): $(1) | $(OBJ)
	@$(CXX) $(CXXFLAGS) -c $$< -o $$@
endef
$(foreach s,$(ALL_SRC),$(eval $(call compile_rule,$(s))))

$(LIB): $(ALL_OBJ)
	@echo "[ref] link $@ ($(words $(ALL_OBJ)) objects)"
	@$(CXX) -shared -pthread -o $@ $(ALL_OBJ) -ldl

$(DRIVER): $(HERE)ref_driver.cpp $(LIB)
	$(CXX) -std=c++11 -O2 -pthread $(INC) $(DEFS) $< -o $@ -L$(OUT) -lduckdb_ref -Wl,-rpath,'$$ORIGIN' -ldl

# the duckdb:: adapter of INTEGRATION.md section 2 (duckdb-polr_amd/host/duckdb_adapter/) run against the reference's own
# JoinHashTable: tests/conformance/adapter_main.cpp, linked against the reference and the device library
ADAPTER := $(OUT)/adapter_test
REPO    := $(abspath $(HERE)..)
adapter: $(ADAPTER)
$(ADAPTER): $(REPO)/tests/conformance/adapter_main.cpp $(REPO)/duckdb-polr_amd/host/duckdb_adapter/polr_duckdb_adapter.hpp $(LIB)
	$(CXX) -std=c++11 -O2 -pthread $(INC) $(DEFS) -I$(REPO)/include -I$(REPO)/duckdb-polr_amd/host $< -o $@ \
	    -L$(OUT) -lduckdb_ref -L$(REPO)/duckdb-polr_amd -lpolr_hip -Wl,-rpath,'$$ORIGIN' -Wl,-rpath,'$$ORIGIN/../../duckdb-polr_amd' -ldl

clean:
	rm -rf $(OUT)
