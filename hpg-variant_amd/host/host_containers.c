/* host_containers.c -- the hpg-libs stand-in containers and records the hot path touches (include/hpgv_host.h).
 * Part of libhpgv_host.so (see hpgv_host_internal.h for the map of its units). */
#include "hpgv_host_internal.h"

/* ------------------------------------------------------------------------ */
/* small containers                                                           */
/* ------------------------------------------------------------------------ */

array_list_t *array_list_new(size_t initial_capacity) {
    array_list_t *l = (array_list_t *)calloc(1, sizeof *l);
    if (!l) return NULL;
    l->capacity = initial_capacity ? initial_capacity : 8;
    l->items = (void **)malloc(l->capacity * sizeof(void *));
    if (!l->items) { free(l); return NULL; }
    return l;
}

int array_list_insert(void *item, array_list_t *list) {
    if (list->size == list->capacity) {
        size_t cap = list->capacity * 2;
        void **p = (void **)realloc(list->items, cap * sizeof(void *));
        if (!p) return 0;
        list->items = p; list->capacity = cap;
    }
    list->items[list->size++] = item;
    return 1;
}

void *array_list_get(size_t index, const array_list_t *list) {
    return index < list->size ? list->items[index] : NULL;
}

void array_list_free(array_list_t *list, void (*item_free)(void *)) {
    if (!list) return;
    if (item_free) for (size_t i = 0; i < list->size; i++) item_free(list->items[i]);
    free(list->items);
    free(list);
}

vcf_record_t *vcf_record_new(void) {
    vcf_record_t *r = (vcf_record_t *)calloc(1, sizeof *r);
    if (r) r->samples = array_list_new(16);
    return r;
}
void vcf_record_free(vcf_record_t *r) { if (r) { array_list_free(r->samples, NULL); free(r); } }
void set_vcf_record_chromosome(char *s, int n, vcf_record_t *r) { r->chromosome = s; r->chromosome_len = n; }
void set_vcf_record_position(long p, vcf_record_t *r) { r->position = (unsigned long)p; }
void set_vcf_record_id(char *s, int n, vcf_record_t *r) { r->id = s; r->id_len = n; }
void set_vcf_record_reference(char *s, int n, vcf_record_t *r) { r->reference = s; r->reference_len = n; }
void set_vcf_record_alternate(char *s, int n, vcf_record_t *r) { r->alternate = s; r->alternate_len = n; }
void set_vcf_record_format(char *s, int n, vcf_record_t *r) { r->format = s; r->format_len = n; }

individual_t *individual_new(char *id, float variable, enum Sex sex, enum Condition condition,
                             individual_t *father, individual_t *mother, struct family *family) {
    individual_t *i = (individual_t *)calloc(1, sizeof *i);
    if (!i) return NULL;
    i->id = id; i->variable = variable; i->sex = sex; i->condition = condition;
    i->father = father; i->mother = mother; i->family = family;
    return i;
}
void individual_free(individual_t *i) { free(i); }

family_t *family_new(char *id) {
    family_t *f = (family_t *)calloc(1, sizeof *f);
    if (!f) return NULL;
    f->id = id;
    f->founders = array_list_new(2);
    f->members = array_list_new(4);
    return f;
}
int family_set_parent(individual_t *p, family_t *f) { return array_list_insert(p, f->founders) ? 0 : 1; }
int family_add_child(individual_t *c, family_t *f) { return array_list_insert(c, f->members) ? 0 : 1; }
void family_free(family_t *f) {
    if (!f) return;
    array_list_free(f->founders, NULL);
    array_list_free(f->members, NULL);
    free(f);
}

static size_t str_hash(const char *s) {
    size_t h = 1469598103934665603ULL;
    for (; *s; s++) { h ^= (unsigned char)*s; h *= 1099511628211ULL; }
    return h;
}

sample_ids_t *sample_ids_new(size_t expected) {
    sample_ids_t *t = (sample_ids_t *)calloc(1, sizeof *t);
    if (!t) return NULL;
    size_t nb = 16;
    while (nb < expected * 2 + 1) nb <<= 1;
    t->n_buckets = nb;
    t->keys = (const char **)calloc(nb, sizeof(char *));
    t->vals = (int *)calloc(nb, sizeof(int));
    if (!t->keys || !t->vals) { free(t->keys); free(t->vals); free(t); return NULL; }
    return t;
}

static int sample_ids_grow(sample_ids_t *t) {
    sample_ids_t *n = sample_ids_new(t->n_buckets);
    if (!n) return 0;
    for (size_t i = 0; i < t->n_buckets; i++)
        if (t->keys[i]) sample_ids_put(n, t->keys[i], t->vals[i]);
    free(t->keys); free(t->vals);
    *t = *n;
    free(n);
    return 1;
}

int sample_ids_put(sample_ids_t *t, const char *name, int position) {
    if ((t->size + 1) * 2 > t->n_buckets && !sample_ids_grow(t)) return 0;
    size_t m = t->n_buckets - 1, i = str_hash(name) & m;
    while (t->keys[i] && strcmp(t->keys[i], name)) i = (i + 1) & m;
    if (!t->keys[i]) { t->keys[i] = name; t->size++; }
    t->vals[i] = position;
    return 1;
}

int sample_ids_get(const sample_ids_t *t, const char *name) {
    size_t m = t->n_buckets - 1, i = str_hash(name) & m;
    while (t->keys[i]) {
        if (!strcmp(t->keys[i], name)) return t->vals[i];
        i = (i + 1) & m;
    }
    return -1;
}

void sample_ids_free(sample_ids_t *t) { if (t) { free(t->keys); free(t->vals); free(t); } }

void list_init(const char *name, int writers, size_t max_length, list_t *list) {
    memset(list, 0, sizeof *list);
    list->name = name ? strdup(name) : NULL;
    list->writers = writers;
    list->max_length = max_length;
    pthread_mutex_init(&list->lock, NULL);
    pthread_cond_init(&list->condition, NULL);
}

list_item_t *list_item_new(int id, int type, void *data_p) {
    list_item_t *it = (list_item_t *)calloc(1, sizeof *it);
    if (it) { it->id = id; it->type = type; it->data_p = data_p; }
    return it;
}
void list_item_free(list_item_t *item) { free(item); }

int list_insert_item(list_item_t *item, list_t *list) {
    if (!item || !list) return 0;
    pthread_mutex_lock(&list->lock);
    item->next_p = NULL;
    if (list->last_p) list->last_p->next_p = item; else list->first_p = item;
    list->last_p = item;
    list->length++;
    pthread_cond_broadcast(&list->condition);
    pthread_mutex_unlock(&list->lock);
    return 1;
}

/* n items chained through next_p, appended in one go: what n list_insert_item calls from one thread leave behind, with one
 * lock and one wake-up instead of n (a consumer asleep on the condition costs a system call per broadcast) */
void list_insert_chain(list_item_t *first, list_item_t *last, size_t n, list_t *list) {
    if (!first || !list) return;
    pthread_mutex_lock(&list->lock);
    last->next_p = NULL;
    if (list->last_p) list->last_p->next_p = first; else list->first_p = first;
    list->last_p = last;
    list->length += n;
    pthread_cond_broadcast(&list->condition);
    pthread_mutex_unlock(&list->lock);
}

list_item_t *list_remove_item(list_t *list) {
    pthread_mutex_lock(&list->lock);
    while (!list->first_p && list->writers > 0) pthread_cond_wait(&list->condition, &list->lock);
    list_item_t *it = list->first_p;
    if (it) {
        list->first_p = it->next_p;
        if (!list->first_p) list->last_p = NULL;
        list->length--;
        it->next_p = NULL;
    }
    pthread_mutex_unlock(&list->lock);
    return it;
}

int list_decr_writers(list_t *list) {
    pthread_mutex_lock(&list->lock);
    if (list->writers > 0) list->writers--;
    pthread_cond_broadcast(&list->condition);
    int w = list->writers;
    pthread_mutex_unlock(&list->lock);
    return w;
}

void list_free_deep(list_t *list, void (*data_free)(void *)) {
    list_item_t *it = list->first_p;
    while (it) {
        list_item_t *n = it->next_p;
        if (data_free && it->data_p) data_free(it->data_p);
        free(it);
        it = n;
    }
    free(list->name);
    pthread_mutex_destroy(&list->lock);
    pthread_cond_destroy(&list->condition);
    memset(list, 0, sizeof *list);
}

void assoc_basic_result_free(assoc_basic_result_t *r) {
    if (!r) return;
    free(r->chromosome); free(r->id); free(r->reference); free(r->alternate); free(r);
}
void assoc_fisher_result_free(assoc_fisher_result_t *r) {
    if (!r) return;
    free(r->chromosome); free(r->id); free(r->reference); free(r->alternate); free(r);
}
void tdt_result_free(tdt_result_t *r) {
    if (!r) return;
    free(r->chromosome); free(r->id); free(r->reference); free(r->alternate); free(r);
}
void variant_stats_free(variant_stats_t *s) {
    if (!s) return;
    free(s->chromosome); free(s->ref_allele); free(s->alt_alleles);
    free(s->alleles_count); free(s->genotypes_count); free(s->alleles_freq); free(s->genotypes_freq);
    free(s->phenotype_stats);
    free(s);
}
sample_stats_t *sample_stats_new(char *name) {
    sample_stats_t *s = (sample_stats_t *)calloc(1, sizeof *s);
    if (s) s->name = name;
    return s;
}
void sample_stats_free(sample_stats_t *s) { free(s); }
file_stats_t *file_stats_new(void) {
    file_stats_t *f = (file_stats_t *)calloc(1, sizeof *f);
    if (f) pthread_mutex_init(&f->lock, NULL);
    return f;
}
void file_stats_free(file_stats_t *f) { if (f) { pthread_mutex_destroy(&f->lock); free(f); } }
