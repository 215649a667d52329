/* ce_pydict.c — CPython helpers of the RLlib vector hook's dict protocol (contracts_amd/vector_env.py).
 *
 * The reference hands RLlib per-agent dictionaries (two_stage_train.py:62-121, cleanup_new.py:258-262); at E = 16 384
 * sub-envs a tick is ~0.4 M dictionary entries.  The hook keeps the dictionary TREES of a tick alive (two generations,
 * recycled alternately) with the observation / feature arrays as views of page-locked snapshot buffers, so a tick only has
 * to touch the entries whose values changed since the tree was last handed out.  These loops do that touching — and the
 * reverse walk over the action dictionaries — in C: no interpreter dispatch per entry, interned keys with cached hashes.
 *
 * Pure host-side marshalling: no env logic, no device code.  Built by contracts_amd/build.py with the system C compiler.
 */
#define PY_SSIZE_T_CLEAN
#include <Python.h>
#include <stdint.h>
#include <string.h>

typedef struct {
  Py_buffer view;
  int held;
} buf_t;

static int get_buf(PyObject* o, buf_t* b, int writable, Py_ssize_t min_bytes, const char* what) {
  b->held = 0;
  if (PyObject_GetBuffer(o, &b->view, writable ? (PyBUF_WRITABLE | PyBUF_C_CONTIGUOUS) : PyBUF_C_CONTIGUOUS) != 0) return -1;
  b->held = 1;
  if (b->view.len < min_bytes) {
    PyErr_Format(PyExc_ValueError, "%s: buffer of %zd bytes, need %zd", what, b->view.len, min_bytes);
    PyBuffer_Release(&b->view);
    b->held = 0;
    return -1;
  }
  return 0;
}
static void put_buf(buf_t* b) {
  if (b->held) PyBuffer_Release(&b->view);
  b->held = 0;
}

/* parse_actions(action_dict, env_keys, agent_keys, out) -> None
 * out[e * n + a] = action_dict[env_keys[e]][agent_keys[a]] as uint8 for every env of env_keys (a list of the E int objects
 * 0 .. E - 1) and every agent key.  KeyError if an env or agent is missing, ValueError for an id outside 0 .. 255. */
static PyObject* parse_actions(PyObject* self, PyObject* args) {
  PyObject *adict, *env_keys, *agent_keys, *out_obj;
  if (!PyArg_ParseTuple(args, "O!O!O!O", &PyDict_Type, &adict, &PyList_Type, &env_keys, &PyTuple_Type, &agent_keys, &out_obj)) return NULL;
  const Py_ssize_t E = PyList_GET_SIZE(env_keys), n = PyTuple_GET_SIZE(agent_keys);
  buf_t out;
  if (get_buf(out_obj, &out, 1, E * n, "parse_actions out") != 0) return NULL;
  uint8_t* dst = (uint8_t*)out.view.buf;
  for (Py_ssize_t e = 0; e < E; ++e) {
    PyObject* ek = PyList_GET_ITEM(env_keys, e);
    PyObject* d = PyDict_GetItemWithError(adict, ek);
    if (d == NULL) {
      if (!PyErr_Occurred()) PyErr_SetObject(PyExc_KeyError, ek);
      goto fail;
    }
    if (!PyDict_Check(d)) {
      PyErr_SetString(PyExc_TypeError, "send_actions: every sub-env's actions must be a dict {agent_id: action}");
      goto fail;
    }
    for (Py_ssize_t a = 0; a < n; ++a) {
      PyObject* ak = PyTuple_GET_ITEM(agent_keys, a);
      PyObject* v = PyDict_GetItemWithError(d, ak);
      if (v == NULL) {
        if (!PyErr_Occurred()) PyErr_SetObject(PyExc_KeyError, ak);
        goto fail;
      }
      long x;
      if (PyLong_CheckExact(v)) {
        x = PyLong_AsLong(v);
      } else {
        PyObject* idx = PyNumber_Index(v); /* numpy integers, 0-d arrays */
        if (idx == NULL) goto fail;
        x = PyLong_AsLong(idx);
        Py_DECREF(idx);
      }
      if (x < 0 || x > 255) {
        if (!PyErr_Occurred()) PyErr_Format(PyExc_ValueError, "action id %ld out of range", x);
        goto fail;
      }
      dst[e * n + a] = (uint8_t)x;
    }
  }
  put_buf(&out);
  Py_RETURN_NONE;
fail:
  put_buf(&out);
  return NULL;
}

/* refresh_floats(dicts, agent_keys, new, shadow) -> entries rewritten
 * dicts: list of E dictionaries {agent_key: float}; new / shadow: float64 [E][n].  Where new differs from shadow (bitwise)
 * the entry becomes float(new) and shadow is brought up to date. */
static PyObject* refresh_floats(PyObject* self, PyObject* args) {
  PyObject *dicts, *agent_keys, *new_obj, *shadow_obj;
  if (!PyArg_ParseTuple(args, "O!O!OO", &PyList_Type, &dicts, &PyTuple_Type, &agent_keys, &new_obj, &shadow_obj)) return NULL;
  const Py_ssize_t E = PyList_GET_SIZE(dicts), n = PyTuple_GET_SIZE(agent_keys);
  buf_t nb, sb;
  if (get_buf(new_obj, &nb, 0, E * n * 8, "refresh_floats new") != 0) return NULL;
  if (get_buf(shadow_obj, &sb, 1, E * n * 8, "refresh_floats shadow") != 0) {
    put_buf(&nb);
    return NULL;
  }
  const uint64_t* nv = (const uint64_t*)nb.view.buf;
  uint64_t* sv = (uint64_t*)sb.view.buf;
  long changed = 0;
  for (Py_ssize_t e = 0; e < E; ++e) {
    const uint64_t *ne = nv + e * n;
    uint64_t* se = sv + e * n;
    if (memcmp(ne, se, (size_t)n * 8) == 0) continue;
    PyObject* d = PyList_GET_ITEM(dicts, e);
    for (Py_ssize_t a = 0; a < n; ++a) {
      if (ne[a] == se[a]) continue;
      double x;
      memcpy(&x, ne + a, 8);
      PyObject* f = PyFloat_FromDouble(x);
      if (f == NULL || PyDict_SetItem(d, PyTuple_GET_ITEM(agent_keys, a), f) != 0) {
        Py_XDECREF(f);
        put_buf(&nb);
        put_buf(&sb);
        return NULL;
      }
      Py_DECREF(f);
      se[a] = ne[a];
      ++changed;
    }
  }
  put_buf(&nb);
  put_buf(&sb);
  return PyLong_FromLong(changed);
}

/* refresh_ints(dicts, agent_keys, new, shadow) -> entries rewritten; as refresh_floats for int32 [E][n] -> int */
static PyObject* refresh_ints(PyObject* self, PyObject* args) {
  PyObject *dicts, *agent_keys, *new_obj, *shadow_obj;
  if (!PyArg_ParseTuple(args, "O!O!OO", &PyList_Type, &dicts, &PyTuple_Type, &agent_keys, &new_obj, &shadow_obj)) return NULL;
  const Py_ssize_t E = PyList_GET_SIZE(dicts), n = PyTuple_GET_SIZE(agent_keys);
  buf_t nb, sb;
  if (get_buf(new_obj, &nb, 0, E * n * 4, "refresh_ints new") != 0) return NULL;
  if (get_buf(shadow_obj, &sb, 1, E * n * 4, "refresh_ints shadow") != 0) {
    put_buf(&nb);
    return NULL;
  }
  const int32_t* nv = (const int32_t*)nb.view.buf;
  int32_t* sv = (int32_t*)sb.view.buf;
  long changed = 0;
  for (Py_ssize_t e = 0; e < E; ++e) {
    const int32_t* ne = nv + e * n;
    int32_t* se = sv + e * n;
    if (memcmp(ne, se, (size_t)n * 4) == 0) continue;
    PyObject* d = PyList_GET_ITEM(dicts, e);
    for (Py_ssize_t a = 0; a < n; ++a) {
      if (ne[a] == se[a]) continue;
      PyObject* f = PyLong_FromLong(ne[a]);
      if (f == NULL || PyDict_SetItem(d, PyTuple_GET_ITEM(agent_keys, a), f) != 0) {
        Py_XDECREF(f);
        put_buf(&nb);
        put_buf(&sb);
        return NULL;
      }
      Py_DECREF(f);
      se[a] = ne[a];
      ++changed;
    }
  }
  put_buf(&nb);
  put_buf(&sb);
  return PyLong_FromLong(changed);
}

/* refresh_infos(agent_dicts, key0, key1, new, shadow) -> entries rewritten
 * agent_dicts: list of E * n per-agent info dictionaries; new / shadow: uint8 [E][n][2]; byte 0 -> d[key0], byte 1 -> d[key1]
 * (key0 may be None: that byte is not part of the dictionary). */
static PyObject* refresh_infos(PyObject* self, PyObject* args) {
  PyObject *dicts, *key0, *key1, *new_obj, *shadow_obj;
  if (!PyArg_ParseTuple(args, "O!OOOO", &PyList_Type, &dicts, &key0, &key1, &new_obj, &shadow_obj)) return NULL;
  const Py_ssize_t N = PyList_GET_SIZE(dicts);
  buf_t nb, sb;
  if (get_buf(new_obj, &nb, 0, N * 2, "refresh_infos new") != 0) return NULL;
  if (get_buf(shadow_obj, &sb, 1, N * 2, "refresh_infos shadow") != 0) {
    put_buf(&nb);
    return NULL;
  }
  const uint16_t* nv = (const uint16_t*)nb.view.buf;
  uint16_t* sv = (uint16_t*)sb.view.buf;
  long changed = 0;
  for (Py_ssize_t i = 0; i < N; ++i) {
    if (nv[i] == sv[i]) continue;
    const uint8_t *nn = (const uint8_t*)(nv + i), *ss = (const uint8_t*)(sv + i);
    PyObject* d = PyList_GET_ITEM(dicts, i);
    for (int k = 0; k < 2; ++k) {
      PyObject* key = k == 0 ? key0 : key1;
      if (nn[k] == ss[k] || key == Py_None) continue;
      PyObject* v = PyLong_FromLong(nn[k]); /* small ints: cached objects, no allocation */
      if (v == NULL || PyDict_SetItem(d, key, v) != 0) {
        Py_XDECREF(v);
        put_buf(&nb);
        put_buf(&sb);
        return NULL;
      }
      Py_DECREF(v);
      ++changed;
    }
    sv[i] = nv[i];
  }
  put_buf(&nb);
  put_buf(&sb);
  return PyLong_FromLong(changed);
}

/* refresh_dones(dicts, keys, new, shadow) -> entries rewritten; dicts: list of E dictionaries, every key of `keys` is set to
 * bool(new[e]) where new[e] differs from shadow[e]; new / shadow: uint8 [E] */
static PyObject* refresh_dones(PyObject* self, PyObject* args) {
  PyObject *dicts, *keys, *new_obj, *shadow_obj;
  if (!PyArg_ParseTuple(args, "O!O!OO", &PyList_Type, &dicts, &PyTuple_Type, &keys, &new_obj, &shadow_obj)) return NULL;
  const Py_ssize_t E = PyList_GET_SIZE(dicts), nk = PyTuple_GET_SIZE(keys);
  buf_t nb, sb;
  if (get_buf(new_obj, &nb, 0, E, "refresh_dones new") != 0) return NULL;
  if (get_buf(shadow_obj, &sb, 1, E, "refresh_dones shadow") != 0) {
    put_buf(&nb);
    return NULL;
  }
  const uint8_t* nv = (const uint8_t*)nb.view.buf;
  uint8_t* sv = (uint8_t*)sb.view.buf;
  long changed = 0;
  for (Py_ssize_t e = 0; e < E; ++e) {
    if (nv[e] == sv[e]) continue;
    PyObject* d = PyList_GET_ITEM(dicts, e);
    PyObject* v = nv[e] ? Py_True : Py_False;
    for (Py_ssize_t k = 0; k < nk; ++k) {
      if (PyDict_SetItem(d, PyTuple_GET_ITEM(keys, k), v) != 0) {
        put_buf(&nb);
        put_buf(&sb);
        return NULL;
      }
    }
    sv[e] = nv[e];
    ++changed;
  }
  put_buf(&nb);
  put_buf(&sb);
  return PyLong_FromLong(changed);
}

/* assign_rows(dicts, keys, rows) -> None; dicts: list of E dictionaries, rows: list of E * len(keys) objects:
 * dicts[e][keys[k]] = rows[e * len(keys) + k] (the Box-space kinds' observation rows, fresh every tick) */
static PyObject* assign_rows(PyObject* self, PyObject* args) {
  PyObject *dicts, *keys, *rows;
  if (!PyArg_ParseTuple(args, "O!O!O!", &PyList_Type, &dicts, &PyTuple_Type, &keys, &PyList_Type, &rows)) return NULL;
  const Py_ssize_t E = PyList_GET_SIZE(dicts), nk = PyTuple_GET_SIZE(keys);
  if (PyList_GET_SIZE(rows) != E * nk) {
    PyErr_SetString(PyExc_ValueError, "assign_rows: rows must hold len(dicts) * len(keys) objects");
    return NULL;
  }
  for (Py_ssize_t e = 0; e < E; ++e) {
    PyObject* d = PyList_GET_ITEM(dicts, e);
    for (Py_ssize_t k = 0; k < nk; ++k)
      if (PyDict_SetItem(d, PyTuple_GET_ITEM(keys, k), PyList_GET_ITEM(rows, e * nk + k)) != 0) return NULL;
  }
  Py_RETURN_NONE;
}

/* assign_key(dicts, key, values) -> None; dicts[i][key] = values[i] (the per-agent info dictionaries' feature_obs rows) */
static PyObject* assign_key(PyObject* self, PyObject* args) {
  PyObject *dicts, *key, *values;
  if (!PyArg_ParseTuple(args, "O!OO!", &PyList_Type, &dicts, &key, &PyList_Type, &values)) return NULL;
  const Py_ssize_t N = PyList_GET_SIZE(dicts);
  if (PyList_GET_SIZE(values) != N) {
    PyErr_SetString(PyExc_ValueError, "assign_key: one value per dictionary");
    return NULL;
  }
  for (Py_ssize_t i = 0; i < N; ++i)
    if (PyDict_SetItem(PyList_GET_ITEM(dicts, i), key, PyList_GET_ITEM(values, i)) != 0) return NULL;
  Py_RETURN_NONE;
}

static PyMethodDef methods[] = {
    {"assign_rows", assign_rows, METH_VARARGS, "dicts[e][keys[k]] = rows[e * len(keys) + k]"},
    {"assign_key", assign_key, METH_VARARGS, "dicts[i][key] = values[i]"},
    {"parse_actions", parse_actions, METH_VARARGS, "action dictionaries -> uint8 [E][n]"},
    {"refresh_floats", refresh_floats, METH_VARARGS, "update {agent: float} dictionaries from a float64 [E][n] snapshot"},
    {"refresh_ints", refresh_ints, METH_VARARGS, "update {agent: int} dictionaries from an int32 [E][n] snapshot"},
    {"refresh_infos", refresh_infos, METH_VARARGS, "update per-agent info dictionaries from a uint8 [E][n][2] snapshot"},
    {"refresh_dones", refresh_dones, METH_VARARGS, "update done dictionaries from a uint8 [E] snapshot"},
    {NULL, NULL, 0, NULL}};

static struct PyModuleDef moddef = {PyModuleDef_HEAD_INIT, "_ce_pydict", "dict-protocol marshalling loops of the vector hook", -1, methods};

PyMODINIT_FUNC PyInit__ce_pydict(void) { return PyModule_Create(&moddef); }
